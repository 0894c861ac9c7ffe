import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig, bf16_bits_to_f32
from oracle import lm_ref
cfg = LMConfig(vocab_size=8192, hidden=512, n_layers=4, n_heads=8, n_kv_heads=2, head_dim=64, ffn=4096)
ids = np.random.default_rng(0).integers(0, 8192, 41)
w = lm_ref.random_weights(cfg, 11, 0.05)
wh = {k: (bf16_bits_to_f32(v).astype(np.float16) if (k.endswith("_proj.weight") or k == "lm_head.weight") else v) for k, v in w.items()}
out = {}
for tag, kw in (("bf16", dict(model_path="random:mid")), ("f16rand", dict(model_path="random:mid", weight_format="f16")), ("f16given", dict(weights=wh)), ("bf16given", dict(weights=w))):
    llm = LlamaForAlternatingCodeChannels(config=cfg, n_ctx=1024, random_seed=11, init_std=0.05, device=0, **kw)
    llm.set_mfma_prefill(False)
    llm.reset(); llm.eval(ids[:1].tolist())
    out[tag] = llm._scores[-1].copy()
    print(tag, llm.weight_format, out[tag][:4])
