"""The long-context parity cases shared by tests/test_lm_long_gpu.py and tests/golden/make_lm_1b_long_golden.py.

Why they exist: the duplex bench runs the decode step at 2.2 k - 8.2 k tokens of context and the reference allocates
n_ctx = 16384 (realtime_agent_resources.py:13; sliding window of realtime_agent_config.py:23-24), while the other LM tests stop
at a few hundred tokens.  Two cases:

  * "wide": the 1B model's LAYER shape (H 2048, 32 q / 8 kv heads x 64, ffn 8192 -- the very GEMV / attention instantiations and
    launch geometry of the bench) with 2 layers and an 8192-token vocabulary, so that the fp32 host oracle can follow it to 16.5 k
    tokens in seconds.  Checkpoints sit on both sides of every context bucket of the step graphs (4 * 2^b splits of 256 keys), of
    the 32-split limit of the in-launch merge (8 kv heads x 32 splits = one workgroup per CU), of the separate combine launch
    behind it, and -- with n_ctx = 20480 -- of the > 64-split tail of the merge.
  * "1b_long": tests/lm_1b_case.py's model (Llama-3.2-1B dims, V = 259 344) behind a 2 200-token context in the duplex grammar.
"""
import numpy as np

# ---------------------------------------------------------------- wide, shallow model followed to 16.5 k tokens
WIDE_SEED, WIDE_STD = 7, 0.02
WIDE_N_CTX = 20480
WIDE_CHECKPOINTS = (1000, 2100, 4200, 8100, 8300, 10000, 16500)
WIDE_TOKENS = WIDE_CHECKPOINTS[-1] + 16


def wide_config():
    from realtime_codec_agent_amd.llm import LMConfig
    return LMConfig(vocab_size=8192, hidden=2048, n_layers=2, n_heads=32, n_kv_heads=8, head_dim=64, ffn=8192)


def wide_ids():
    return np.random.default_rng(77).integers(0, 8192, WIDE_TOKENS).astype(np.int64)


# ---------------------------------------------------------------- the 1B model behind 2 200 tokens
LONG_1B_CTX = 2200


def long_1b_ids():
    """24 text ids, then [agent, user] codec-id pairs (the grammar of realtime_agent_v2.py:84-99), then two S = 2 steps"""
    rng = np.random.default_rng(4242)
    text = rng.integers(0, 128256, 24)
    codec = rng.integers(128266, 128266 + 131072, LONG_1B_CTX - 24 + 4)
    ids = np.concatenate([text, codec]).astype(np.int64)
    return ids[:LONG_1B_CTX], [ids[LONG_1B_CTX:LONG_1B_CTX + 2], ids[LONG_1B_CTX + 2:LONG_1B_CTX + 4]]


def long_1b_oracle_points(weight_format=None):
    """LMRef (fp16 KV) over the case: [last-token logits after the context, after step 1, after step 2]"""
    import torch

    import lm_1b_case as case
    from oracle import lm_ref
    cfg = case.config()
    ctx, steps = long_1b_ids()
    used = np.concatenate([ctx] + steps)
    w = lm_ref.random_weights(cfg, case.SEED, case.INIT_STD, embed_rows=used)
    if weight_format in ("q8_0", "q4_k"):
        from oracle import q4k_ref, q8_ref
        fq = q8_ref.fake_quant if weight_format == "q8_0" else q4k_ref.fake_quant
        for k in list(w):
            if k.endswith("_proj.weight") or k == "lm_head.weight":
                w[k] = fq(w[k])
    ref = lm_ref.LMRef(cfg, w, kv_dtype=torch.float16)
    pts = [ref.eval(ctx, last_only=True, chunk=512)[-1].numpy()]
    for s in steps:
        pts.append(ref.eval(s, last_only=True)[-1].numpy())
    return pts
