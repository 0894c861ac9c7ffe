"""Test helper: writes a llama-architecture GGUF v3 file the way convert_hf_to_gguf.py lays one out (metadata keys,
reversed dims, Q/K row permutation, 32-byte aligned tensor data) so the importer can be exercised offline."""
import struct

import numpy as np

F32, F16, Q8_0, Q4_K, Q6_K, BF16 = 0, 1, 8, 12, 14, 30


def _s(b: bytes) -> bytes:
    return struct.pack("<Q", len(b)) + b


def _kv(key: str, vtype: int, payload: bytes) -> bytes:
    return _s(key.encode()) + struct.pack("<I", vtype) + payload


def permute(w: np.ndarray, n_head: int) -> np.ndarray:
    """convert_hf_to_gguf.py LlamaModel.permute (rotate-half rows -> interleaved pairs)."""
    return w.reshape(n_head, 2, w.shape[0] // n_head // 2, *w.shape[1:]).swapaxes(1, 2).reshape(w.shape)


def quantize(a: np.ndarray, ttype: int) -> bytes:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if ttype == F32:
        return a.tobytes()
    if ttype == F16:
        return a.astype(np.float16).tobytes()
    if ttype == BF16:
        u = a.view(np.uint32)
        return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16).tobytes()
    if ttype == Q8_0:
        blk = a.reshape(-1, 32)
        d = np.abs(blk).max(axis=1, keepdims=True) / 127.0
        q = np.where(d > 0, np.round(blk / np.where(d > 0, d, 1.0)), 0).astype(np.int8)
        out = np.empty((blk.shape[0], 34), np.uint8)
        out[:, :2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
        out[:, 2:] = q.view(np.uint8)
        return out.tobytes()
    if ttype == Q4_K:      # 144-byte super-blocks, scales chosen by this repo's simple min / max rule (oracle/q4k_ref.py)
        import os, sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import q4k_ref
        return q4k_ref.pack_blocks(q4k_ref.quantize_q4_k(a.reshape(-1, 256))).tobytes()
    if ttype == Q6_K:      # 210-byte super-blocks
        import os, sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from oracle import q4k_ref
        return q4k_ref.pack_blocks_q6_k(q4k_ref.quantize_q6_k(a.reshape(-1, 256))).tobytes()
    raise ValueError(ttype)


def q4_k_m_type(gguf_name: str, n_layers: int) -> int:
    """The mix llama-quantize writes for LLAMA_FTYPE_MOSTLY_Q4_K_M (llama.cpp llama_tensor_get_type): output.weight Q6_K; attn_v and
    ffn_down Q6_K in the layers use_more_bits() picks (first and last eighth, every third in between), Q4_K otherwise."""
    if gguf_name == "output.weight":
        return Q6_K
    if ".attn_v." in gguf_name or ".ffn_down." in gguf_name:
        i = int(gguf_name.split(".")[1])
        more = i < n_layers // 8 or i >= 7 * n_layers // 8 or (i - n_layers // 8) % 3 == 2
        return Q6_K if more else Q4_K
    return Q4_K


def write_llama_gguf(path, cfg, weights, matrix_type=F32, tokens=None, rope_freqs=None):
    """cfg: LMConfig; weights: HF-named float32 arrays.  Matrices get `matrix_type` (or, for matrix_type="Q4_K_M", llama-quantize's
    Q4_K / Q6_K mix), norms stay F32 (as llama.cpp does)."""
    mix = matrix_type == "Q4_K_M"
    if mix:
        matrix_type = Q4_K
    kv = []
    kv.append(_kv("general.architecture", 8, _s(b"llama")))
    kv.append(_kv("general.alignment", 4, struct.pack("<I", 32)))
    for key, val in (("llama.embedding_length", cfg.hidden), ("llama.block_count", cfg.n_layers), ("llama.attention.head_count", cfg.n_heads),
                     ("llama.attention.head_count_kv", cfg.n_kv_heads), ("llama.feed_forward_length", cfg.ffn),
                     ("llama.rope.dimension_count", cfg.head_dim), ("llama.context_length", 2048)):
        kv.append(_kv(key, 4, struct.pack("<I", val)))
    kv.append(_kv("llama.attention.layer_norm_rms_epsilon", 6, struct.pack("<f", cfg.rms_eps)))
    kv.append(_kv("llama.rope.freq_base", 6, struct.pack("<f", cfg.rope_theta)))
    if tokens is not None:
        kv.append(_kv("tokenizer.ggml.tokens", 9, struct.pack("<IQ", 8, len(tokens)) + b"".join(_s(t.encode()) for t in tokens)))
    ts = [("token_embd.weight", weights["model.embed_tokens.weight"], matrix_type), ("output_norm.weight", weights["model.norm.weight"], F32),
          ("output.weight", weights["lm_head.weight"], matrix_type)]
    if rope_freqs is not None:
        ts.append(("rope_freqs.weight", np.asarray(rope_freqs, np.float32), F32))
    g = {"self_attn.q_proj": "attn_q", "self_attn.k_proj": "attn_k", "self_attn.v_proj": "attn_v", "self_attn.o_proj": "attn_output",
         "mlp.gate_proj": "ffn_gate", "mlp.up_proj": "ffn_up", "mlp.down_proj": "ffn_down", "input_layernorm": "attn_norm",
         "post_attention_layernorm": "ffn_norm"}
    for l in range(cfg.n_layers):
        for hf, gg in g.items():
            a = np.asarray(weights[f"model.layers.{l}.{hf}.weight"], np.float32)
            if gg == "attn_q":
                a = permute(a, cfg.n_heads)
            elif gg == "attn_k":
                a = permute(a, cfg.n_kv_heads)
            ts.append((f"blk.{l}.{gg}.weight", a, F32 if a.ndim == 1 else matrix_type))
    if mix:
        ts = [(name, a, tt if (tt == F32 or a.ndim == 1) else q4_k_m_type(name, cfg.n_layers)) for name, a, tt in ts]
    infos, blobs, off = [], [], 0
    for name, a, tt in ts:
        data = quantize(a, tt)
        ne = list(reversed(a.shape))
        infos.append(_s(name.encode()) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", d) for d in ne) + struct.pack("<IQ", tt, off))
        pad = (-len(data)) % 32
        blobs.append(data + b"\0" * pad)
        off += len(data) + pad
    head = struct.pack("<IIQQ", 0x46554747, 3, len(ts), len(kv)) + b"".join(kv) + b"".join(infos)
    with open(path, "wb") as f:
        f.write(head + b"\0" * ((-len(head)) % 32))
        for b in blobs:
            f.write(b)
