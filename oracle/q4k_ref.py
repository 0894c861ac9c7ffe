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


# ------------------------------------------------------------------------------------------------------------------ Q6_K
# `Q4_K_M` is a MIX: llama-quantize writes output.weight and (in the layers its use_more_bits() picks) attn_v / ffn_down as Q6_K.
# ggml-common.h `block_q6_K` (210 bytes, 256 weights) = { uint8 ql[128]; uint8 qh[64]; int8 scales[16]; fp16 d }, and
# ggml-quants.c `dequantize_row_q6_K`: for each half n of 128 weights (ql += 64, qh += 32, scales += 8) and l = 0..31, is = l / 16:
#     q1 = ((ql[l]      & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32    -> y[l]      = d * scales[is + 0] * q1
#     q2 = ((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32    -> y[l + 32] = d * scales[is + 2] * q2
#     q3 = ((ql[l]      >> 4)  | (((qh[l] >> 4) & 3) << 4)) - 32    -> y[l + 64] = d * scales[is + 4] * q3
#     q4 = ((ql[l + 32] >> 4)  | (((qh[l] >> 6) & 3) << 4)) - 32    -> y[l + 96] = d * scales[is + 6] * q4
# i.e. weight k of a block has the 6-bit value q_k - 32, the int8 scale of its group of 16 and the block's fp16 d; d * scale * q is
# exact in f32 (11 + 7 + 6 bits).
def quantize_q6_k(w: np.ndarray):
    """float32 [..., K] (K % 256 == 0) -> dict(q int8 [..., K] in -32..31, sc int8 [..., K/16], d float16 [..., K/256]).  This build's
    simple symmetric rule (llama-quantize searches with make_qx_quants)."""
    w = np.ascontiguousarray(w, dtype=np.float32)
    lead, K = w.shape[:-1], w.shape[-1]
    assert K % 256 == 0
    g = w.reshape(-1, K // 256, 16, 16)
    amax = np.abs(g).max(axis=-1)                                        # per group of 16
    s = (amax / np.float32(31.0)).astype(np.float32)                     # group step
    d = (s.max(axis=-1) / np.float32(127.0)).astype(np.float16)
    df = d.astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(df > 0, _round_half_up(s / np.where(df > 0, df, 1)), 0)
    sc = np.clip(sc, 0, 127).astype(np.float32)
    step = (df * sc).astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(step > 0, np.floor(g / np.where(step > 0, step, 1) + np.float32(0.5)), 0)
    q = np.clip(q, -32, 31).astype(np.int8)
    return dict(q=q.reshape(*lead, K), sc=sc.astype(np.int8).reshape(*lead, K // 16), d=d.reshape(*lead, K // 256))


def pack_blocks_q6_k(p: dict) -> np.ndarray:
    """-> raw GGUF block_q6_K bytes uint8 [n_blocks, 210]"""
    u = (p["q"].reshape(-1, 2, 4, 32).astype(np.int16) + 32).astype(np.uint8)        # [blk][half n][quarter 0..3][l]; weight 128 n + 32 quarter + l
    nb = u.shape[0]
    lo, hi = u & 0xF, u >> 4
    ql = np.empty((nb, 2, 64), np.uint8)
    ql[:, :, 0:32] = lo[:, :, 0] | (lo[:, :, 2] << 4)          # ql[l]      : low nibble q1 (quarter 0), high nibble q3 (quarter 2)
    ql[:, :, 32:64] = lo[:, :, 1] | (lo[:, :, 3] << 4)         # ql[l + 32] : low nibble q2 (quarter 1), high nibble q4 (quarter 3)
    qh = hi[:, :, 0] | (hi[:, :, 1] << 2) | (hi[:, :, 2] << 4) | (hi[:, :, 3] << 6)   # [nb, 2, 32]
    out = np.empty((nb, 210), np.uint8)
    out[:, 0:128] = ql.reshape(nb, 128)
    out[:, 128:192] = qh.reshape(nb, 64)
    out[:, 192:208] = p["sc"].reshape(nb, 16).view(np.uint8)
    out[:, 208:210] = p["d"].reshape(-1, 1).view(np.uint8)
    return out


def dequantize_blocks_q6_k(raw: np.ndarray) -> np.ndarray:
    """raw blocks uint8 [nb, 210] -> float32 [nb * 256] by dequantize_row_q6_K's rule"""
    raw = raw.reshape(-1, 210)
    nb = raw.shape[0]
    ql = raw[:, 0:128].reshape(nb, 2, 64)
    qh = raw[:, 128:192].reshape(nb, 2, 32)
    sc = raw[:, 192:208].copy().view(np.int8).astype(np.float32).reshape(nb, 2, 8)       # per half: scales[is + 0 / 2 / 4 / 6], is = l / 16
    d = raw[:, 208:210].copy().view(np.float16).astype(np.float32).reshape(nb, 1, 1)
    q = np.empty((nb, 2, 4, 32), np.int16)
    q[:, :, 0] = (ql[:, :, 0:32] & 0xF) | (((qh >> 0) & 3) << 4)
    q[:, :, 1] = (ql[:, :, 32:64] & 0xF) | (((qh >> 2) & 3) << 4)
    q[:, :, 2] = (ql[:, :, 0:32] >> 4) | (((qh >> 4) & 3) << 4)
    q[:, :, 3] = (ql[:, :, 32:64] >> 4) | (((qh >> 6) & 3) << 4)
    q = (q - 32).astype(np.float32)
    # quarter j, l -> scale index is + 2 j with is = l / 16
    scq = np.stack([np.repeat(sc[:, :, 2 * j:2 * j + 2], 16, axis=-1) for j in range(4)], axis=2)     # [nb, 2, 4, 32]
    return ((d[..., None] * scq).astype(np.float32) * q).astype(np.float32).reshape(-1)


def fake_quant_q6_k(w: np.ndarray) -> np.ndarray:
    if w.dtype == np.uint16:
        w = (w.astype(np.uint32) << 16).view(np.float32)
    return dequantize_blocks_q6_k(pack_blocks_q6_k(quantize_q6_k(w))).reshape(w.shape)
