import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import lm_ref, q8_ref
from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
cfg = LMConfig(vocab_size=8192, hidden=512, n_layers=int(os.environ.get("NL", "4")), n_heads=8, n_kv_heads=2, head_dim=64, ffn=4096)
llm = LlamaForAlternatingCodeChannels(model_path="random:mid", config=cfg, n_ctx=1024, random_seed=11, init_std=0.05, device=0, weight_format="q8_0")
llm.set_mfma_prefill(False)
ref = lm_ref.LMRef(cfg, q8_ref.quantized_model(lm_ref.random_weights(cfg, 11, 0.05)), kv_dtype=torch.float16)
ids = np.random.default_rng(0).integers(0, 8192, 8)
for n in (1, 2):
    llm.reset(); ref.reset()
    llm.eval(ids[:n].tolist()); want = ref.eval(ids[:n])[-1].numpy()
    got = llm._scores[-1]
    print(f"q8 M={n}: nan {np.isnan(got).sum()} / {got.size}  max|d| {np.nanmax(np.abs(got - want)):.3e}", flush=True)
    llm.set_q8_decode(False)
    llm.reset(); llm.eval(ids[:n].tolist()); got2 = llm._scores[-1]
    print(f"bf16(d*q) M={n}: nan {np.isnan(got2).sum()}  max|d| vs oracle {np.nanmax(np.abs(got2 - want)):.3e}", flush=True)
    llm.set_q8_decode(True)
