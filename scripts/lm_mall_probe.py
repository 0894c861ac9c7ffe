"""How much of a decode layer's time is HBM streaming vs fixed latency?  Times the graph-replayed 2-token step for
models of 1/2/4/16 layers with a small head: the 1->2 layer increment has its weights resident in the 256 MB
Infinity Cache (122 MB per layer), the 4->16 increment streams from HBM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig

ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
res = {}
for nl in (1, 2, 4, 16):
    cfg = LMConfig(vocab_size=8192, n_layers=nl)
    llm = LlamaForAlternatingCodeChannels(model_path="random:probe", config=cfg, n_ctx=4096, device=0)
    llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=42)
    ids = np.random.default_rng(0).integers(0, 8192, ctx + 2).tolist()
    llm.eval(ids[:ctx]); llm.sync()
    for _ in range(5):
        llm.n_tokens = ctx; llm.step(ids[ctx:ctx + 2])
    llm.sync(); t0 = time.perf_counter()
    for _ in range(200):
        llm.n_tokens = ctx; llm.step(ids[ctx:ctx + 2])
    llm.sync(); res[nl] = (time.perf_counter() - t0) / 200 * 1e6
    print(f"layers={nl}: step {res[nl]:.1f} us")
    del llm
print(f"per-layer, weights cache-resident (2-1): {res[2]-res[1]:.1f} us; (4-2)/2: {(res[4]-res[2])/2:.1f} us; HBM-streamed (16-4)/12: {(res[16]-res[4])/12:.1f} us")
