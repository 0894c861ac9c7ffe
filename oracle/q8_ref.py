"""GGUF Q8_0 on the host (TEST INFRASTRUCTURE ONLY): the quantisation rule of llama.cpp's quantize_row_q8_0 -- what `llama-quantize
... q8_0` applies when the reference builds its deployed file (prep_test_model.sh:29; ggml is a third-party dependency absent from
/root/reference, so the published rule is restated): per block of 32 values d = max|x| / 127, q = round(x / d) (ties away from
zero), d stored as fp16; value = d * q.  The HIP library quantises on the device with the same rule (lm_q8_quantize_kernel) and its
decode GEMV multiplies d * (sum q x) in f32; the oracle path is LMRef over the de-quantised matrices."""
import numpy as np


def quantize_q8_0(w: np.ndarray):
    """float32 [..., K] (K % 32 == 0) -> (q int8 [..., K], d float16 [..., K / 32])"""
    w = np.ascontiguousarray(w, dtype=np.float32)
    blk = w.reshape(-1, 32)
    amax = np.abs(blk).max(axis=1)
    d = (amax / np.float32(127.0)).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    x = blk * inv[:, None]
    q = (np.sign(x) * np.floor(np.abs(x) + np.float32(0.5))).astype(np.int8)
    return q.reshape(w.shape), d.astype(np.float16).reshape(*w.shape[:-1], w.shape[-1] // 32)


def dequantize_q8_0(q: np.ndarray, d: np.ndarray) -> np.ndarray:
    return (q.astype(np.float32).reshape(*d.shape, 32) * d.astype(np.float32)[..., None]).reshape(q.shape)


def fake_quant(w: np.ndarray) -> np.ndarray:
    """bf16 bits or float32 matrix -> float32 d * q"""
    if w.dtype == np.uint16:
        w = (w.astype(np.uint32) << 16).view(np.float32)
    return dequantize_q8_0(*quantize_q8_0(w))


def quantized_model(weights: dict) -> dict:
    """The model the device runs with weight_format='q8_0': every projection matrix and lm_head replaced by d * q (f32);
    embedding table and norms unchanged."""
    out = {}
    for k, v in weights.items():
        is_proj = k.endswith("_proj.weight") or k == "lm_head.weight"
        out[k] = fake_quant(v) if is_proj else v
    return out
