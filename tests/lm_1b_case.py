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


def oracle_points():
    """LMRef (fp16 KV, like the HIP cache) over the case: [logits after the context, after step 1, after step 2]."""
    import torch
    from oracle import lm_ref
    cfg = config()
    ctx, steps = token_ids()
    used = np.concatenate([ctx] + steps)
    ref = lm_ref.LMRef(cfg, lm_ref.random_weights(cfg, SEED, INIT_STD, embed_rows=used), kv_dtype=torch.float16)
    pts = [ref.eval(ctx)[-1].numpy()]
    for s in steps:
        pts.append(ref.eval(s)[-1].numpy())
    return pts
