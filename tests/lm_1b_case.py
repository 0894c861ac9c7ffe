"""The ~1B parity case shared by tests/golden/make_lm_1b_golden.py (CPU oracle -> committed fixture) and
tests/test_lm_gpu.py (HIP vs the fixture and vs the oracle run live): Llama-3.2-1B dims with the codec vocabulary
(BASELINE config 3), weights from the device's counter hash (seed 0, std 0.02), a 96-token
context in the duplex grammar -- text ids, then [agent, user] codec-id pairs -- followed by two S=2 steps."""
import numpy as np

SEED, INIT_STD = 0, 0.02
N_CTX_TOKENS = 96
TOPK = 100
STRIDE = 61          # logits[::61] -> 4252 values per point


def config():
    from realtime_codec_agent_amd.llm import LMConfig
    return LMConfig.llama_3_2_1b()


def token_ids():
    rng = np.random.default_rng(2024)
    text = rng.integers(0, 128256, 24)
    codec = rng.integers(128266, 128266 + 131072, N_CTX_TOKENS - 24 + 4)
    ids = np.concatenate([text, codec]).astype(np.int64)
    return ids[:N_CTX_TOKENS], [ids[N_CTX_TOKENS:N_CTX_TOKENS + 2], ids[N_CTX_TOKENS + 2:N_CTX_TOKENS + 4]]


def summarize(logits: np.ndarray) -> dict:
    logits = np.asarray(logits, np.float32)
    order = np.argsort(-logits, kind="stable")[:TOPK]
    return dict(top_ids=order.astype(np.int64), top_vals=logits[order].copy(), strided=logits[::STRIDE].copy(),
                mean=np.float64(logits.astype(np.float64).mean()), std=np.float64(logits.astype(np.float64).std()))


def oracle_points(weight_format=None):
    """LMRef (fp16 KV, like the HIP cache) over the case: [logits after the context, after step 1, after step 2].
    weight_format="q8_0": over the model whose projections and lm_head went through llama.cpp's q8_0 rule (oracle/q8_ref.py);
    "q4_k": through this build's Q4_K quantiser and llama.cpp's dequantize_row_q4_K (oracle/q4k_ref.py)."""
    import torch
    from oracle import lm_ref
    cfg = config()
    ctx, steps = token_ids()
    used = np.concatenate([ctx] + steps)
    w = lm_ref.random_weights(cfg, SEED, INIT_STD, embed_rows=used)
    if weight_format in ("q8_0", "q4_k"):
        from oracle import q4k_ref, q8_ref
        fq = q8_ref.fake_quant if weight_format == "q8_0" else q4k_ref.fake_quant
        for k in list(w):
            if k.endswith("_proj.weight") or k == "lm_head.weight":
                w[k] = fq(w[k])       # one matrix at a time: the 1B lm_head alone is 2 GB in f32
    ref = lm_ref.LMRef(cfg, w, kv_dtype=torch.float16)
    pts = [ref.eval(ctx)[-1].numpy()]
    for s in steps:
        pts.append(ref.eval(s)[-1].numpy())
    return pts
