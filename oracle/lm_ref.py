"""Plain PyTorch fp32 restatement of the LM step (TEST INFRASTRUCTURE ONLY).

Follows the arithmetic of the reference's training-time twin, CodecLlamaForCausalLM
(realtime_codec_agent/codec_llama.py:93-164), i.e. transformers' Llama decoder stack:
RMSNorm -> q/k/v proj -> RoPE (rotate_half convention, optional llama3 frequency scaling) ->
causal GQA attention over a KV cache -> o_proj + residual -> RMSNorm -> SwiGLU MLP + residual ->
final RMSNorm -> lm_head.  After persist_codec_embeddings (codec_llama.py:178-206) the input
embedding is a plain table lookup, which is the deployed form restated here.

PINNED: tests/golden/lm_tiny.npz holds logits produced by the reference's own classes
(tests/golden/make_lm_golden.py); test_lm_cpu.py checks this restatement against them.

`kv_dtype=torch.float16` reproduces what the HIP path stores in its cache (llama.cpp's default cache
type); `kv_dtype=None` keeps K/V in fp32 as HF does.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Optional

import numpy as np
import torch

from . import build


def inv_freq(cfg) -> torch.Tensor:
    dim = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(dtype=torch.float) / dim))
    if cfg.rope_scaling == "llama3":
        factor, low, high, old = cfg.rope_factor, cfg.rope_low_freq_factor, cfg.rope_high_freq_factor, cfg.rope_orig_ctx
        low_wl, high_wl = old / low, old / high
        wl = 2 * math.pi / inv
        inv_l = torch.where(wl > low_wl, inv / factor, inv)
        smooth = (old / wl - low) / (high - low)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        medium = ~(wl < high_wl) * ~(wl > low_wl)
        inv = torch.where(medium, smoothed, inv_l)
    return inv.float()


def _rotate_half(x):
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


class LMRef:
    def __init__(self, cfg, weights: Dict[str, np.ndarray], kv_dtype: Optional[torch.dtype] = torch.float16):
        """weights: name -> float32 array, or uint16 array of bf16 bit patterns."""
        self.cfg = cfg
        self.kv_dtype = kv_dtype
        self.w = {}
        for k, v in weights.items():
            if isinstance(v, _SparseRows):
                self.w[k] = v
                continue
            if v.dtype == np.uint16:       # bf16 bits: widened by the C helper (a 1B-size lm_head is 0.5 G values)
                f = np.empty(v.shape, np.float32)
                v = np.ascontiguousarray(v)
                _slib().oracle_bf16_to_f32(v.ctypes.data_as(C.POINTER(C.c_uint16)), C.c_int64(v.size), f.ctypes.data_as(C.POINTER(C.c_float)))
                v = f
            self.w[k] = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
        self.inv_freq = inv_freq(cfg)
        self.reset()

    def reset(self):
        self.n_tokens = 0
        self.k: List[Optional[torch.Tensor]] = [None] * self.cfg.n_layers
        self.v: List[Optional[torch.Tensor]] = [None] * self.cfg.n_layers

    def set_n_tokens(self, n: int):
        """KV rollback: keep the first n cached positions."""
        self.n_tokens = n
        for l in range(self.cfg.n_layers):
            if self.k[l] is not None:
                self.k[l] = self.k[l][:, :n]
                self.v[l] = self.v[l][:, :n]

    def _norm(self, x, w):
        var = x.pow(2).mean(-1, keepdim=True)
        return w * (x * torch.rsqrt(var + self.cfg.rms_eps))

    @torch.no_grad()
    def eval(self, ids, last_only: bool = False, chunk: Optional[int] = None, drop_keys=None) -> torch.Tensor:
        """Append ids at position n_tokens; returns logits [len(ids), V] (fp32), or [1, V] of the last position with last_only
        (a long context at a 259 k vocabulary: the head product of every position is TFLOPs nobody reads).  chunk: walk a long
        eval in pieces of that many tokens (bounds the [heads, S, T] score tensor; same arithmetic per row).  drop_keys=(a, b): keys
        [a, b) are masked out -- ONLY for the "this tolerance would catch a lost attention split" assertions of the long-context tests."""
        ids = list(ids)
        if chunk and len(ids) > chunk:
            outs = []
            for i in range(0, len(ids), chunk):
                last = i + chunk >= len(ids)
                o = self._eval(ids[i:i + chunk], last_only, drop_keys)
                if not last_only or last:
                    outs.append(o)
            return torch.cat(outs, 0) if not last_only else outs[-1]
        return self._eval(ids, last_only, drop_keys)

    def _eval(self, ids, last_only, drop_keys) -> torch.Tensor:
        c = self.cfg
        ids = torch.as_tensor(list(ids), dtype=torch.long)
        S = ids.shape[0]
        pos = torch.arange(self.n_tokens, self.n_tokens + S)
        freqs = pos[:, None].float() * self.inv_freq[None, :]
        emb = torch.cat((freqs, freqs), dim=-1)
        cos, sin = emb.cos()[None], emb.sin()[None]  # [1,S,hd]
        x = self.w["model.embed_tokens.weight"][ids]  # [S,H]
        G = c.n_heads // c.n_kv_heads
        for l in range(c.n_layers):
            p = f"model.layers.{l}."
            h = self._norm(x, self.w[p + "input_layernorm.weight"])
            q = (h @ self.w[p + "self_attn.q_proj.weight"].T).view(S, c.n_heads, c.head_dim).transpose(0, 1)
            k = (h @ self.w[p + "self_attn.k_proj.weight"].T).view(S, c.n_kv_heads, c.head_dim).transpose(0, 1)
            v = (h @ self.w[p + "self_attn.v_proj.weight"].T).view(S, c.n_kv_heads, c.head_dim).transpose(0, 1)
            q = q * cos + _rotate_half(q) * sin
            k = k * cos + _rotate_half(k) * sin
            if self.kv_dtype is not None:
                k, v = k.to(self.kv_dtype).float(), v.to(self.kv_dtype).float()
            if self.k[l] is not None and self.n_tokens > 0:
                k = torch.cat((self.k[l][:, : self.n_tokens], k), dim=1)
                v = torch.cat((self.v[l][:, : self.n_tokens], v), dim=1)
            self.k[l], self.v[l] = k, v
            T = k.shape[1]
            # GQA without materialising the repeated K / V (a 10 k-token cache times G is gigabytes): [nkv, G * S, hd] queries
            qg = q.reshape(c.n_kv_heads, G * S, c.head_dim)
            att = (qg @ k.transpose(1, 2)) * (c.head_dim ** -0.5)  # [nkv, G*S, T]
            mask = torch.arange(T)[None, :] > (self.n_tokens + torch.arange(S))[:, None]
            if drop_keys is not None:
                mask = mask | ((torch.arange(T) >= drop_keys[0]) & (torch.arange(T) < drop_keys[1]))[None, :]
            att = att.view(c.n_kv_heads, G, S, T).masked_fill(mask[None, None], float("-inf")).softmax(-1).view(c.n_kv_heads, G * S, T)
            o = (att @ v).view(c.n_heads, S, c.head_dim).transpose(0, 1).reshape(S, c.n_heads * c.head_dim)
            x = x + o @ self.w[p + "self_attn.o_proj.weight"].T
            h = self._norm(x, self.w[p + "post_attention_layernorm.weight"])
            g = h @ self.w[p + "mlp.gate_proj.weight"].T
            u = h @ self.w[p + "mlp.up_proj.weight"].T
            x = x + (torch.nn.functional.silu(g) * u) @ self.w[p + "mlp.down_proj.weight"].T
        if last_only:
            x = x[-1:]
        x = self._norm(x, self.w["model.norm.weight"])
        self.n_tokens += S
        return x @ self.w["lm_head.weight"].T


# ------------------------------------------------------------------ random-init weights (bench configs)
_M64 = (1 << 64) - 1


def _splitmix_np(seed: int, ctr: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + ctr * np.uint64(0xBF58476D1CE4E5B9) + np.uint64(0x94D049BB133111EB))
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z


def random_bf16_tensor_c(seed: int, tensor_id: int, n: int, init_std: float, start: int = 0) -> np.ndarray:
    """random_bf16_tensor through oracle/lm_init_oracle.c (OpenMP): the 1.5 G weights of the ~1B bench model in seconds."""
    scale = np.float32(np.float32(init_std) * np.float32(1.7320508) / np.float32(65535.0))
    out = np.empty(n, np.uint16)
    _slib().oracle_random_bf16(C.c_uint64(seed), C.c_uint64(tensor_id), C.c_int64(start), C.c_int64(n), C.c_float(float(scale)),
                               out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def random_bf16_tensor(seed: int, tensor_id: int, n: int, init_std: float, start: int = 0) -> np.ndarray:
    """Same values as lm_random_bf16_kernel (rca_lm.hip): Irwin-Hall sum of four 16-bit uniforms of a
    splitmix64 hash, scaled in float32, rounded to bf16 (RNE).  Returns uint16 bf16 bits of elements
    [start, start + n) of the tensor; generated in blocks so that the 1B-model tensors fit in memory."""
    with np.errstate(over="ignore"):
        s = int(np.uint64(seed) ^ (np.uint64(tensor_id) * np.uint64(0xD6E8FEB86659FD93)))
    scale = np.float32(np.float32(init_std) * np.float32(1.7320508) / np.float32(65535.0))
    m = np.uint64(0xFFFF)
    out = np.empty(n, np.uint16)
    BLK = 1 << 24
    for b0 in range(0, n, BLK):
        b1 = min(n, b0 + BLK)
        z = _splitmix_np(s, np.arange(start + b0, start + b1, dtype=np.uint64))
        total = ((z & m) + ((z >> np.uint64(16)) & m) + ((z >> np.uint64(32)) & m) + ((z >> np.uint64(48)) & m)).astype(np.int64)
        f = (total - 131070).astype(np.float32) * scale
        u = f.view(np.uint32)
        out[b0:b1] = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
    return out


def random_embedding_rows(cfg, seed: int, ids, init_std: float = 0.02) -> np.ndarray:
    """Rows `ids` of the random-init embedding table (tensor id 1) without generating all V of them."""
    H = cfg.hidden
    return np.stack([random_bf16_tensor(seed, 1, H, init_std, start=int(i) * H) for i in ids])


class _SparseRows:
    """Stand-in for a [V, H] float table of which only a few rows exist: indexing with a LongTensor of ids returns them."""

    def __init__(self, rows: Dict[int, np.ndarray], H: int):
        self.rows, self.H = rows, H
        self.dtype = np.float32

    def __getitem__(self, ids):
        return torch.stack([torch.from_numpy(self.rows[int(i)]) for i in ids])


def _embed_table(cfg, seed, init_std, embed_rows):
    V, H = cfg.vocab_size, cfg.hidden
    if embed_rows is None:
        return random_bf16_tensor_c(seed, 1, V * H, init_std).reshape(V, H)
    rows = {}
    for i in set(int(t) for t in embed_rows):
        b = random_bf16_tensor_c(seed, 1, H, init_std, start=i * H)
        rows[i] = (b.astype(np.uint32) << 16).view(np.float32)
    return _SparseRows(rows, H)


def random_weights(cfg, seed: int, init_std: float = 0.02, embed_rows=None) -> Dict[str, np.ndarray]:
    """HF-named state dict equal to what rca_lm_create_random generates (fused qkv / interleaved
    gate-up tensors are split back into their HF parts).  embed_rows: when given, only those rows of the embedding
    table are generated (the others stay zero): a 1B-size table is 0.5 G values of which a test reads a few dozen."""
    random_bf16_tensor = random_bf16_tensor_c
    H, V, F = cfg.hidden, cfg.vocab_size, cfg.ffn
    Q, KVD = cfg.n_heads * cfg.head_dim, cfg.n_kv_heads * cfg.head_dim
    w = {
        "model.embed_tokens.weight": _embed_table(cfg, seed, init_std, embed_rows),
        "lm_head.weight": random_bf16_tensor(seed, 2, V * H, init_std).reshape(V, H),
        "model.norm.weight": np.ones(H, np.float32),
    }
    for l in range(cfg.n_layers):
        p = f"model.layers.{l}."
        base = 10 * (l + 1)
        qkv = random_bf16_tensor(seed, base + 3, (Q + 2 * KVD) * H, init_std).reshape(Q + 2 * KVD, H)
        w[p + "self_attn.q_proj.weight"], w[p + "self_attn.k_proj.weight"], w[p + "self_attn.v_proj.weight"] = qkv[:Q], qkv[Q:Q + KVD], qkv[Q + KVD:]
        w[p + "self_attn.o_proj.weight"] = random_bf16_tensor(seed, base + 4, H * Q, init_std).reshape(H, Q)
        gu = random_bf16_tensor(seed, base + 5, 2 * F * H, init_std).reshape(F, 2, H)
        w[p + "mlp.gate_proj.weight"], w[p + "mlp.up_proj.weight"] = np.ascontiguousarray(gu[:, 0]), np.ascontiguousarray(gu[:, 1])
        w[p + "mlp.down_proj.weight"] = random_bf16_tensor(seed, base + 6, H * F, init_std).reshape(H, F)
        w[p + "input_layernorm.weight"] = np.ones(H, np.float32)
        w[p + "post_attention_layernorm.weight"] = np.ones(H, np.float32)
    return w


# ------------------------------------------------------------------ codec-embedding projector
def project_codec_embeddings(codec_embed: np.ndarray, w1: np.ndarray, b1: np.ndarray, w2: np.ndarray, b2: np.ndarray) -> np.ndarray:
    """Rows that CodecLlamaForCausalLM.persist_codec_embeddings writes into the table (codec_llama.py:178-206):
    linear_2(gelu(linear_1(e))) with the exact erf GELU (CodecLlamaMultiModalProjector, codec_llama.py:32-44), fp32."""
    e = torch.from_numpy(np.asarray(codec_embed, np.float32))
    h = torch.nn.functional.linear(e, torch.from_numpy(np.asarray(w1, np.float32)), torch.from_numpy(np.asarray(b1, np.float32)))
    h = torch.nn.functional.gelu(h)
    return torch.nn.functional.linear(h, torch.from_numpy(np.asarray(w2, np.float32)), torch.from_numpy(np.asarray(b2, np.float32))).numpy()


# ------------------------------------------------------------------ sampler (C restatement)
_lib = None


def _slib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_sample.restype = C.c_int
        _lib.oracle_sample_ex.restype = C.c_int
        _lib.oracle_expf.restype = C.c_float
        _lib.oracle_expf.argtypes = [C.c_float]
        _lib.oracle_logf.restype = C.c_float
        _lib.oracle_logf.argtypes = [C.c_float]
    return _lib


def sample(logits: np.ndarray, top_k: int, top_p: float, min_p: float, temp: float, seed: int, counter: int,
           logit_bias: Optional[Dict[int, float]] = None, repeat_penalty: float = 1.0, frequency_penalty: float = 0.0,
           presence_penalty: float = 0.0, prev_tokens=()) -> int:
    """prev_tokens: the tokens this sampler accepted before this draw (the restatement looks at the last 64, like llama.cpp's
    penalties sampler with llama-cpp-python's default window)."""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    bias = logit_bias or {}
    ids = (C.c_int32 * max(1, len(bias)))(*bias.keys())
    vals = (C.c_float * max(1, len(bias)))(*bias.values())
    prev = [int(t) for t in list(prev_tokens)[-64:]]
    pv = (C.c_int32 * max(1, len(prev)))(*prev)
    return _slib().oracle_sample_ex(logits.ctypes.data_as(C.POINTER(C.c_float)), logits.shape[0], int(top_k), C.c_float(top_p), C.c_float(min_p),
                                    C.c_float(temp), C.c_uint32(seed & 0xFFFFFFFF), C.c_uint64(counter), len(bias), ids, vals,
                                    C.c_float(repeat_penalty), C.c_float(frequency_penalty), C.c_float(presence_penalty), len(prev), pv)


def expf(x: float) -> float:
    return float(_slib().oracle_expf(C.c_float(x)))


def logf(x: float) -> float:
    return float(_slib().oracle_logf(C.c_float(x)))
