"""GGUF Q4_K on the host (TEST INFRASTRUCTURE ONLY).

`llama-quantize ... Q4_K_M` is the third file the reference deploys (prep_test_model.sh:31).  ggml is a third-party dependency that is
absent from /root/reference, so the published format and de-quantisation rule are restated here (ggml-common.h `block_q4_K`,
ggml-quants.c `dequantize_row_q4_K` / `get_scale_min_k4`):

    block_q4_K (144 bytes, 256 weights) = { fp16 d; fp16 dmin; uint8 scales[12]; uint8 qs[128] }
    sub-block j (32 weights, j = 0..7) has a 6-bit scale sc_j and a 6-bit minimum m_j packed into scales[12]:
        j < 4 : sc_j = scales[j] & 63                                   m_j = scales[j + 4] & 63
        j >= 4: sc_j = (scales[j + 4] & 0xF) | ((scales[j - 4] >> 6) << 4)    m_j = (scales[j + 4] >> 4) | ((scales[j] >> 6) << 4)
    weights 64 t .. 64 t + 31 are the LOW nibbles of qs[32 t .. 32 t + 31] (sub-block 2 t), weights 64 t + 32 .. 64 t + 63 the HIGH
    nibbles of the same bytes (sub-block 2 t + 1)
    value = (d * sc_j) * q - (dmin * m_j)         (f32: two products, one subtraction, in this order)

What is pinned to llama.cpp is that rule.  The QUANTISER below is this build's own simple min / max rule (llama-quantize searches for
better scales with make_qkx2_quants; any block it writes de-quantises by the rule above, which is all the device path depends on):
the HIP library applies the same rule on the device for `weight_format="q4_k"` (lm_q4k_quantize_kernel), block for block.
"""
import numpy as np


def _round_half_up(x):
    return np.floor(x + np.float32(0.5))


def quantize_q4_k(w: np.ndarray):
    """float32 [..., K] (K % 256 == 0) -> dict(q uint8 [..., K] in 0..15, sc uint8 [..., K/32], m uint8 [..., K/32],
    d float16 [..., K/256], dmin float16 [..., K/256])."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    lead, K = w.shape[:-1], w.shape[-1]
    assert K % 256 == 0
    sub = w.reshape(-1, K // 256, 8, 32)
    mn = np.minimum(sub.min(axis=-1), np.float32(0.0))                      # <= 0
    mx = sub.max(axis=-1)
    s = ((mx - mn) / np.float32(15.0)).astype(np.float32)                   # sub-block step
    o = (-mn).astype(np.float32)                                            # sub-block offset >= 0
    d = (s.max(axis=-1) / np.float32(63.0)).astype(np.float16)
    dmin = (o.max(axis=-1) / np.float32(63.0)).astype(np.float16)
    df, dminf = d.astype(np.float32)[..., None], dmin.astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(df > 0, _round_half_up(s / np.where(df > 0, df, 1)), 0)
        m = np.where(dminf > 0, _round_half_up(o / np.where(dminf > 0, dminf, 1)), 0)
    sc = np.clip(sc, 0, 63).astype(np.float32)
    m = np.clip(m, 0, 63).astype(np.float32)
    d1 = (df * sc).astype(np.float32)[..., None]
    m1 = (dminf * m).astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(d1 > 0, _round_half_up((sub + m1) / np.where(d1 > 0, d1, 1)), 0)
    q = np.clip(q, 0, 15).astype(np.uint8)
    return dict(q=q.reshape(*lead, K), sc=sc.astype(np.uint8).reshape(*lead, K // 32), m=m.astype(np.uint8).reshape(*lead, K // 32),
                d=d.reshape(*lead, K // 256), dmin=dmin.reshape(*lead, K // 256))


def pack_blocks(p: dict) -> np.ndarray:
    """the dict of quantize_q4_k -> raw GGUF blocks uint8 [n_blocks, 144]"""
    q = p["q"].reshape(-1, 4, 2, 32)                     # [blk][t][low / high][32]
    sc = p["sc"].reshape(-1, 8).astype(np.uint8)
    m = p["m"].reshape(-1, 8).astype(np.uint8)
    nb = q.shape[0]
    out = np.empty((nb, 144), np.uint8)
    out[:, 0:2] = p["d"].reshape(-1, 1).view(np.uint8)
    out[:, 2:4] = p["dmin"].reshape(-1, 1).view(np.uint8)
    scales = np.zeros((nb, 12), np.uint8)
    scales[:, 0:4] = (sc[:, 0:4] & 63) | ((sc[:, 4:8] >> 4) << 6)
    scales[:, 4:8] = (m[:, 0:4] & 63) | ((m[:, 4:8] >> 4) << 6)
    scales[:, 8:12] = (sc[:, 4:8] & 0xF) | ((m[:, 4:8] & 0xF) << 4)
    out[:, 4:16] = scales
    out[:, 16:144] = (q[:, :, 0, :] | (q[:, :, 1, :] << 4)).reshape(nb, 128)
    return out


def unpack_scales(scales: np.ndarray):
    """get_scale_min_k4 for all eight sub-blocks: uint8 [nb, 12] -> (sc uint8 [nb, 8], m uint8 [nb, 8])"""
    sc = np.empty(scales.shape[:-1] + (8,), np.uint8)
    m = np.empty_like(sc)
    sc[..., 0:4] = scales[..., 0:4] & 63
    m[..., 0:4] = scales[..., 4:8] & 63
    sc[..., 4:8] = (scales[..., 8:12] & 0xF) | ((scales[..., 0:4] >> 6) << 4)
    m[..., 4:8] = (scales[..., 8:12] >> 4) | ((scales[..., 4:8] >> 6) << 4)
    return sc, m


def dequantize_blocks(raw: np.ndarray) -> np.ndarray:
    """raw blocks uint8 [nb, 144] -> float32 [nb * 256] by dequantize_row_q4_K's rule"""
    raw = raw.reshape(-1, 144)
    d = raw[:, 0:2].copy().view(np.float16).astype(np.float32)          # [nb, 1]
    dmin = raw[:, 2:4].copy().view(np.float16).astype(np.float32)
    sc, m = unpack_scales(raw[:, 4:16])
    qs = raw[:, 16:144].reshape(-1, 4, 32)
    q = np.stack([qs & 0xF, qs >> 4], axis=2).astype(np.float32)         # [nb, 4, 2, 32] -> sub-block 2 t + {0, 1}
    d1 = (d * sc.astype(np.float32)).astype(np.float32).reshape(-1, 4, 2, 1)
    m1 = (dmin * m.astype(np.float32)).astype(np.float32).reshape(-1, 4, 2, 1)
    return ((d1 * q).astype(np.float32) - m1).astype(np.float32).reshape(-1)


def fake_quant(w: np.ndarray) -> np.ndarray:
    """bf16 bits or float32 matrix [N, K] -> float32 values of its Q4_K blocks (this build's quantiser, GGUF's de-quantiser)"""
    if w.dtype == np.uint16:
        w = (w.astype(np.uint32) << 16).view(np.float32)
    return dequantize_blocks(pack_blocks(quantize_q4_k(w))).reshape(w.shape)


def quantized_model(weights: dict) -> dict:
    """The model the device runs with weight_format='q4_k': every projection matrix and lm_head replaced by its Q4_K values (f32);
    embedding table and norms unchanged."""
    out = {}
    for k, v in weights.items():
        is_proj = k.endswith("_proj.weight") or k == "lm_head.weight"
        out[k] = fake_quant(v) if is_proj else v
    return out
