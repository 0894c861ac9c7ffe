"""LM step timing at a given context length (graph replay vs eager). Usage: lm_profile.py [ctx] [steps] [eager]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
eager = len(sys.argv) > 3 and sys.argv[3] == "eager"
cfg = LMConfig.llama_3_2_1b()
fmt = os.environ.get("RCA_LM_FORMAT")   # "q8_0": decode from packed q8_0 weights
llm = LlamaForAlternatingCodeChannels(model_path="random:1b", config=cfg, n_ctx=16384, device=0, weight_format=fmt)
llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=42)
rng = np.random.default_rng(0)
ids = rng.integers(128266, 259338, ctx + 2).tolist()
llm.eval(ids[:ctx]); llm.sync()          # cold pass (module load, first-touch)
llm.reset()
t0 = time.perf_counter(); llm.eval(ids[:ctx]); llm.sync(); t_prefill = time.perf_counter() - t0
llm.set_graphs(not eager)
for _ in range(5):
    llm.n_tokens = ctx; llm.step(ids[ctx:ctx + 2])
llm.sync(); t0 = time.perf_counter()
for _ in range(steps):
    llm.n_tokens = ctx; llm.step(ids[ctx:ctx + 2])
llm.sync(); dt = (time.perf_counter() - t0) / steps
wb = llm.weight_bytes_per_step(); kv = 2 * 2 * cfg.n_layers * cfg.n_kv_heads * cfg.head_dim * ctx
print(f"fmt={llm.weight_format} ctx={ctx} mode={'eager' if eager else 'graph'} step_ms={dt*1e3:.3f} GB/s={(wb+kv)/dt/1e9:.0f} prefill_s={t_prefill:.3f} ({ctx/t_prefill:.0f} tok/s)")
