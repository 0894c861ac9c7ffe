"""LM leg of __graft_entry__.smoke(): a few fused steps of a small random-init model on cuda:0, checked
against the torch oracle (logits) and the C sampler restatement (token)."""
import numpy as np


def run() -> None:
    import torch
    from oracle import lm_ref
    from realtime_codec_agent_amd.llm import LlamaForAlternatingCodeChannels, LMConfig
    cfg = LMConfig(vocab_size=4096, hidden=256, n_layers=2, n_heads=4, n_kv_heads=1, head_dim=64, ffn=512)
    llm = LlamaForAlternatingCodeChannels(model_path="random:smoke", config=cfg, n_ctx=256, random_seed=5, init_std=0.05, device=0)
    ref = lm_ref.LMRef(cfg, lm_ref.random_weights(cfg, 5, 0.05), kv_dtype=torch.float16)
    llm.init_sampler_for_generate(top_k=100, top_p=1.0, min_p=0.0, temp=1.0, seed=42)
    ids = np.random.default_rng(0).integers(0, 4096, 12).tolist()
    llm.eval(ids[:10])
    ref.eval(ids[:10])
    tok = llm.step(ids[10:12])
    want_logits = ref.eval(ids[10:12])[-1].numpy()
    got = llm._scores[-1]
    d = float(np.abs(got - want_logits).max())
    assert d < 1e-3, f"LM logits differ from the oracle by {d}"
    assert tok == lm_ref.sample(got, 100, 1.0, 0.0, 1.0, 42, 0), "sampled token differs from the restatement"
    print(f"smoke lm ok: token {tok}, max|dlogit| {d:.2e}")
