"""GGUF import for the LM.

The reference runs its LM from a GGUF file through llama.cpp (realtime_agent_resources.py:12,19-25;
prep_test_model.sh:28-31 produces F16 / Q8_0 / Q4_K_M files with convert_hf_to_gguf.py + llama-quantize).
This module reads such a file -- format v2/v3, little endian, architecture "llama" -- and returns what
`llm.load_weights` returns for a safetensors directory: an LMConfig and tensors under their Hugging Face
names, so a GGUF path can be handed to LlamaForAlternatingCodeChannels / RealtimeAgentResources unchanged.

Supported tensor types: F32 (stored as bf16 on the device), BF16, F16 (2-D tensors stay fp16, RCA_F16: the reference's default file
is the F16 one, realtime_agent_resources.py:12) and Q8_0: the projection matrices and output.weight of a Q8_0 file are handed to
the library as their raw 34-byte blocks (RCA_Q8_0) and stay packed in HBM, one copy, for decode and prefill; the embedding table is
de-quantised exactly (f32 rows on the device: llama.cpp's get_rows does the same per looked-up row).  K-quants (Q4_K_M ...) are
rejected with a clear error -- except the two a Q4_K_M file (prep_test_model.sh:31) is made of: Q4_K (144-byte super-blocks, kept packed,
RCA_Q4_K) and Q6_K (210-byte super-blocks: output.weight and the attn_v / ffn_down tensors llama-quantize's use_more_bits() picks;
RCA_Q6_K, re-encoded losslessly on the device as int8 values + one f32 scale per 16).

Two things convert_hf_to_gguf.py does to a Llama checkpoint are undone here:
  * q_proj / k_proj rows are permuted from the rotate-half layout to interleaved pairs (LlamaModel.permute);
    the kernels use the Hugging Face layout, so the inverse permutation is applied;
  * llama3 rope scaling is stored as a `rope_freqs.weight` tensor of per-frequency divisors; it is returned as
    "rope.inv_freq" so the device RoPE table uses exactly the file's frequencies.
The file's tokenizer vocabulary (tokenizer.ggml.tokens) is returned in the metadata for building the tokenizer.
"""
from __future__ import annotations

import struct
from typing import Any, BinaryIO, Dict, Tuple

import numpy as np

GGUF_MAGIC = 0x46554747  # "GGUF"
GGML_F32, GGML_F16, GGML_Q8_0, GGML_Q4_K, GGML_Q6_K, GGML_BF16 = 0, 1, 8, 12, 14, 30
_TYPE_NAMES = {0: "F32", 1: "F16", 2: "Q4_0", 3: "Q4_1", 6: "Q5_0", 7: "Q5_1", 8: "Q8_0", 9: "Q8_1", 10: "Q2_K", 11: "Q3_K",
               12: "Q4_K", 13: "Q5_K", 14: "Q6_K", 15: "Q8_K", 30: "BF16"}
# metadata value types
_U8, _I8, _U16, _I16, _U32, _I32, _F32, _BOOL, _STR, _ARR, _U64, _I64, _F64 = range(13)
_SCALAR = {_U8: "<B", _I8: "<b", _U16: "<H", _I16: "<h", _U32: "<I", _I32: "<i", _F32: "<f", _BOOL: "<?", _U64: "<Q", _I64: "<q", _F64: "<d"}
_NP = {_U8: np.uint8, _I8: np.int8, _U16: np.uint16, _I16: np.int16, _U32: np.uint32, _I32: np.int32, _F32: np.float32,
       _BOOL: np.bool_, _U64: np.uint64, _I64: np.int64, _F64: np.float64}


class GGUFError(ValueError):
    pass


def _read(f: BinaryIO, fmt: str):
    n = struct.calcsize(fmt)
    b = f.read(n)
    if len(b) != n:
        raise GGUFError("unexpected end of file")
    return struct.unpack(fmt, b)[0]


def _read_str(f: BinaryIO) -> str:
    n = _read(f, "<Q")
    if n > (1 << 28):
        raise GGUFError(f"implausible string length {n}")
    return f.read(n).decode("utf-8", errors="replace")


def _read_value(f: BinaryIO, t: int):
    if t in _SCALAR:
        return _read(f, _SCALAR[t])
    if t == _STR:
        return _read_str(f)
    if t == _ARR:
        et = _read(f, "<I")
        n = _read(f, "<Q")
        if et in _NP:
            a = np.frombuffer(f.read(n * np.dtype(_NP[et]).itemsize), dtype=_NP[et])
            if len(a) != n:
                raise GGUFError("unexpected end of file in array")
            return a
        if et == _STR:
            return [_read_str(f) for _ in range(n)]
        return [_read_value(f, et) for _ in range(n)]
    raise GGUFError(f"unknown metadata value type {t}")


def _dequant(raw: np.ndarray, ttype: int, numel: int) -> np.ndarray:
    """-> float32 [numel] (or uint16 bf16 bits for BF16 / float16 for F16, which the loader uploads as they are)."""
    if ttype == GGML_F32:
        return raw.view(np.float32)[:numel]
    if ttype == GGML_F16:
        return raw.view(np.float16)[:numel]
    if ttype == GGML_BF16:
        return raw.view(np.uint16)[:numel]
    if ttype == GGML_Q8_0:  # blocks of 32: f16 scale + 32 x int8
        nb = numel // 32
        blk = raw[: nb * 34].reshape(nb, 34)
        d = blk[:, :2].copy().view(np.float16).astype(np.float32)          # [nb,1]
        q = blk[:, 2:].view(np.int8).astype(np.float32)                    # [nb,32]
        return (q * d).reshape(-1)
    if ttype == GGML_Q4_K:
        from ._native import Q4KBlocks
        return Q4KBlocks(raw[: numel // 256 * 144], (numel // 256, 256)).dequantize().reshape(-1)
    if ttype == GGML_Q6_K:
        from ._native import Q6KBlocks
        return Q6KBlocks(raw[: numel // 256 * 210], (numel // 256, 256)).dequantize().reshape(-1)
    raise GGUFError(f"tensor type {_TYPE_NAMES.get(ttype, ttype)} is not supported (F32, F16, BF16, Q8_0, Q4_K, Q6_K are)")


def _nbytes(ttype: int, numel: int) -> int:
    if ttype == GGML_F32:
        return 4 * numel
    if ttype in (GGML_F16, GGML_BF16):
        return 2 * numel
    if ttype == GGML_Q8_0:
        if numel % 32:
            raise GGUFError("Q8_0 tensor whose size is not a multiple of 32")
        return numel // 32 * 34
    if ttype == GGML_Q4_K:
        if numel % 256:
            raise GGUFError("Q4_K tensor whose size is not a multiple of 256")
        return numel // 256 * 144
    if ttype == GGML_Q6_K:
        if numel % 256:
            raise GGUFError("Q6_K tensor whose size is not a multiple of 256")
        return numel // 256 * 210
    raise GGUFError(f"tensor type {_TYPE_NAMES.get(ttype, ttype)} is not supported (F32, F16, BF16, Q8_0, Q4_K, Q6_K are: the types of the "
                    "reference's F16 / Q8_0 / Q4_K_M files)")


def read_gguf(path: str, keep_q8_0: bool = False) -> Tuple[Dict[str, Any], Dict[str, np.ndarray]]:
    """-> (metadata, tensors).  Tensors keep their GGUF names; shapes are row-major (reversed `ne`).  keep_q8_0: 2-D Q8_0 tensors
    come back as _native.Q8Blocks (the raw 34-byte blocks) instead of being de-quantised."""
    with open(path, "rb") as f:
        if _read(f, "<I") != GGUF_MAGIC:
            raise GGUFError(f"{path}: not a GGUF file")
        version = _read(f, "<I")
        if version not in (2, 3):
            raise GGUFError(f"{path}: GGUF version {version} is not supported (2 and 3 are)")
        n_tensors, n_kv = _read(f, "<Q"), _read(f, "<Q")
        meta: Dict[str, Any] = {"gguf.version": version}
        for _ in range(n_kv):
            key = _read_str(f)
            meta[key] = _read_value(f, _read(f, "<I"))
        infos = []
        for _ in range(n_tensors):
            name = _read_str(f)
            nd = _read(f, "<I")
            ne = [_read(f, "<Q") for _ in range(nd)]
            ttype = _read(f, "<I")
            off = _read(f, "<Q")
            infos.append((name, ne, ttype, off))
        align = int(meta.get("general.alignment", 32))
        base = (f.tell() + align - 1) // align * align
        mm = np.memmap(path, dtype=np.uint8, mode="r")
        tensors: Dict[str, np.ndarray] = {}
        for name, ne, ttype, off in infos:
            numel = int(np.prod(ne)) if ne else 1
            nb = _nbytes(ttype, numel)
            if base + off + nb > mm.shape[0]:
                raise GGUFError(f"tensor '{name}' runs past the end of the file")
            raw = np.asarray(mm[base + off: base + off + nb])
            if keep_q8_0 and ttype == GGML_Q8_0 and len(ne) == 2 and ne[0] % 32 == 0:
                from ._native import Q8Blocks
                tensors[name] = Q8Blocks(raw, tuple(reversed(ne)))
                continue
            if keep_q8_0 and ttype == GGML_Q4_K and len(ne) == 2 and ne[0] % 256 == 0:
                from ._native import Q4KBlocks
                tensors[name] = Q4KBlocks(raw, tuple(reversed(ne)))
                continue
            if keep_q8_0 and ttype == GGML_Q6_K and len(ne) == 2 and ne[0] % 256 == 0:
                from ._native import Q6KBlocks
                tensors[name] = Q6KBlocks(raw, tuple(reversed(ne)))
                continue
            tensors[name] = _dequant(raw, ttype, numel).reshape(tuple(reversed(ne)) if ne else ())
        return meta, tensors


def _unpermute(w, n_head: int):
    """Inverse of convert_hf_to_gguf.py LlamaModel.permute: rows [head][hd/2][2] -> [head][2][hd/2].  A Q8_0 tensor kept as blocks
    is permuted row-wise on its raw bytes (a block never crosses a row)."""
    rows = w.shape[0]
    if hasattr(w, "take_rows"):
        idx = np.arange(rows).reshape(n_head, rows // n_head // 2, 2).swapaxes(1, 2).reshape(-1)
        return w.take_rows(idx)
    return w.reshape(n_head, rows // n_head // 2, 2, *w.shape[1:]).swapaxes(1, 2).reshape(w.shape)


def load_llama_gguf(path: str, keep_q8_0: bool = True):
    """-> (LMConfig, {HF tensor name: ndarray}, metadata) for a llama-architecture GGUF.  Q8_0 projection matrices and output.weight
    stay packed (Q8Blocks): the device re-lays the blocks out and the decode step streams them as 8.5 bits per weight; everything
    else (embedding table, norms) is de-quantised here."""
    from .llm import LMConfig, rope_inv_freq
    meta, t = read_gguf(path, keep_q8_0=keep_q8_0)
    tied_head = t.get("token_embd.weight")          # a file without output.weight ties lm_head to the table: the head keeps the blocks
    if hasattr(tied_head, "dequantize"):
        t["token_embd.weight"] = tied_head.dequantize()   # exact f32 rows (the device keeps the table in f32)
    arch = meta.get("general.architecture", "llama")
    if arch != "llama":
        raise GGUFError(f"{path}: architecture '{arch}' is not supported (llama is)")

    def m(key, default=None):
        v = meta.get(f"{arch}.{key}", default)
        if v is None:
            raise GGUFError(f"{path}: metadata key {arch}.{key} is missing")
        return v

    hidden, n_layers, n_heads = int(m("embedding_length")), int(m("block_count")), int(m("attention.head_count"))
    n_kv = int(m("attention.head_count_kv", n_heads))
    head_dim = int(meta.get(f"{arch}.rope.dimension_count", hidden // n_heads))
    embd = t["token_embd.weight"]
    cfg = LMConfig(vocab_size=int(embd.shape[0]), hidden=hidden, n_layers=n_layers, n_heads=n_heads, n_kv_heads=n_kv, head_dim=head_dim,
                   ffn=int(m("feed_forward_length")), rms_eps=float(m("attention.layer_norm_rms_epsilon", 1e-5)),
                   rope_theta=float(m("rope.freq_base", 10000.0)), rope_scaling=None)
    w: Dict[str, np.ndarray] = {"model.embed_tokens.weight": embd, "model.norm.weight": t["output_norm.weight"],
                                "lm_head.weight": t.get("output.weight", tied_head)}
    names = {"attn_q": "self_attn.q_proj", "attn_k": "self_attn.k_proj", "attn_v": "self_attn.v_proj", "attn_output": "self_attn.o_proj",
             "ffn_gate": "mlp.gate_proj", "ffn_up": "mlp.up_proj", "ffn_down": "mlp.down_proj", "attn_norm": "input_layernorm",
             "ffn_norm": "post_attention_layernorm"}
    for l in range(n_layers):
        for g, hf in names.items():
            key = f"blk.{l}.{g}.weight"
            if key not in t:
                raise GGUFError(f"{path}: tensor {key} is missing")
            a = t[key]
            if g == "attn_q":
                a = _unpermute(a, n_heads)
            elif g == "attn_k":
                a = _unpermute(a, n_kv)
            w[f"model.layers.{l}.{hf}.weight"] = a
    for k, v in list(w.items()):       # 1-D tensors (norm weights) go up as f32
        if isinstance(v, np.ndarray) and v.dtype == np.float16 and v.ndim < 2:
            w[k] = v.astype(np.float32)
    inv = rope_inv_freq(cfg)
    if "rope_freqs.weight" in t:  # llama3 scaling: per-frequency divisors computed by the converter
        inv = (inv / np.asarray(t["rope_freqs.weight"], dtype=np.float32).reshape(-1)).astype(np.float32)
    w["rope.inv_freq"] = inv
    return cfg, w, meta
