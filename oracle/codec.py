"""ctypes front-end of oracle/codec_oracle.c (TEST INFRASTRUCTURE ONLY).

OracleCodec restates AudioTokenizer._magicodec_encode / _magicodec_decode
(reference audio_tokenizer.py:189-201) on the CPU with a fixed fma order.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np

from . import build

MAX_STAGES = 8


class _Cfg(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_int32),
        ("n_stages", C.c_int32),
        ("strides", C.c_int32 * MAX_STAGES),
        ("channels", C.c_int32 * (MAX_STAGES + 1)),
        ("k_in", C.c_int32),
        ("k_latent", C.c_int32),
        ("latent_dim", C.c_int32),
        ("codebook_size", C.c_int32),
        ("codebook_raw_dim", C.c_int32),
        ("codebook_dim", C.c_int32),
        ("leaky_slope", C.c_float),
    ]


_FP = C.POINTER(C.c_float)


class _Weights(C.Structure):
    _fields_ = [
        ("enc_in_w", _FP), ("enc_in_b", _FP),
        ("enc_down_w", _FP * MAX_STAGES), ("enc_down_b", _FP * MAX_STAGES),
        ("enc_out_w", _FP), ("enc_out_b", _FP),
        ("q_in_w", _FP), ("q_in_b", _FP),
        ("q_codebook", _FP),
        ("q_proj_w", _FP), ("q_proj_b", _FP),
        ("dec_in_w", _FP), ("dec_in_b", _FP),
        ("dec_up_w", _FP * MAX_STAGES), ("dec_up_b", _FP * MAX_STAGES),
        ("dec_out_w", _FP), ("dec_out_b", _FP),
    ]


def _fp(a: np.ndarray):
    return a.ctypes.data_as(_FP)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.oracle_codec_encode.restype = C.c_int
        _lib.oracle_codec_decode.restype = C.c_int
    return _lib


def make_cfg(cfg) -> _Cfg:
    c = _Cfg()
    c.sample_rate = cfg.sample_rate
    c.n_stages = len(cfg.strides)
    for i, s in enumerate(cfg.strides):
        c.strides[i] = s
    for i, ch in enumerate(cfg.channels):
        c.channels[i] = ch
    c.k_in, c.k_latent, c.latent_dim = cfg.k_in, cfg.k_latent, cfg.latent_dim
    c.codebook_size, c.codebook_raw_dim, c.codebook_dim = cfg.codebook_size, cfg.codebook_raw_dim, cfg.codebook_dim
    c.leaky_slope = cfg.leaky_slope
    return c


class OracleCodec:
    def __init__(self, cfg, weights: Dict[str, np.ndarray]):
        self.cfg = cfg
        self._c = make_cfg(cfg)
        # keep contiguous float32 copies alive
        self._w = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items()}
        w = _Weights()
        g = self._w
        w.enc_in_w, w.enc_in_b = _fp(g["enc.conv_in.weight"]), _fp(g["enc.conv_in.bias"])
        n = len(cfg.strides)
        for i in range(n):
            w.enc_down_w[i], w.enc_down_b[i] = _fp(g[f"enc.down.{i}.weight"]), _fp(g[f"enc.down.{i}.bias"])
            w.dec_up_w[i], w.dec_up_b[i] = _fp(g[f"dec.up.{i}.weight"]), _fp(g[f"dec.up.{i}.bias"])
        w.enc_out_w, w.enc_out_b = _fp(g["enc.conv_out.weight"]), _fp(g["enc.conv_out.bias"])
        w.q_in_w, w.q_in_b = _fp(g["quantizer.in_proj.weight"]), _fp(g["quantizer.in_proj.bias"])
        w.q_codebook = _fp(g["quantizer.codebook.weight"])
        w.q_proj_w, w.q_proj_b = _fp(g["quantizer.codebook_proj.weight"]), _fp(g["quantizer.codebook_proj.bias"])
        w.dec_in_w, w.dec_in_b = _fp(g["dec.conv_in.weight"]), _fp(g["dec.conv_in.bias"])
        w.dec_out_w, w.dec_out_b = _fp(g["dec.conv_out.weight"]), _fp(g["dec.conv_out.bias"])
        self._wstruct = w
        N, J = cfg.codebook_size, cfg.codebook_dim
        self.cb = np.empty((N, J), np.float32)
        self.hc = np.empty((N,), np.float32)
        lib().oracle_codebook(C.byref(self._c), w.q_codebook, w.q_proj_w, w.q_proj_b, _fp(self.cb), _fp(self.hc))

    @property
    def hop(self) -> int:
        return int(np.prod(self.cfg.strides))

    def codebook(self) -> np.ndarray:
        return self.cb

    def num_frames(self, T: int) -> int:
        return (T + self.hop - 1) // self.hop

    def encode(self, pcm: np.ndarray, tap_layer: int = -1):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        assert pcm.ndim == 2
        B, T = pcm.shape
        F = self.num_frames(T)
        codes = np.empty((B, F), np.int64)
        tap = None
        tap_ptr = None
        if tap_layer >= 0:
            tap = np.empty(self._tap_shape(tap_layer, B, F), np.float32)
            tap_ptr = _fp(tap)
        r = lib().oracle_codec_encode(C.byref(self._c), C.byref(self._wstruct), _fp(self.cb), _fp(self.hc), _fp(pcm),
                                      B, T, codes.ctypes.data_as(C.POINTER(C.c_int64)), tap_layer, tap_ptr)
        assert r == F
        return (codes, tap) if tap_layer >= 0 else codes

    def _tap_shape(self, layer: int, B: int, F: int):
        cfg = self.cfg
        n = len(cfg.strides)
        L = F * self.hop
        if layer == 0:
            return (B, cfg.channels[0], L)
        if 1 <= layer <= n:
            for i in range(layer):
                L //= cfg.strides[i]
            return (B, cfg.channels[layer], L)
        if layer == n + 1:
            return (B, cfg.latent_dim, F)
        if layer == n + 2:
            return (B * F, cfg.codebook_dim)
        raise ValueError(layer)

    def decode(self, codes: np.ndarray) -> np.ndarray:
        codes = np.ascontiguousarray(codes, dtype=np.int64)
        assert codes.ndim == 2
        B, F = codes.shape
        pcm = np.empty((B, F * self.hop), np.float32)
        r = lib().oracle_codec_decode(C.byref(self._c), C.byref(self._wstruct), _fp(self.cb),
                                      codes.ctypes.data_as(C.POINTER(C.c_int64)), B, F, _fp(pcm))
        if r < 0:
            raise ValueError("code out of range")
        return pcm

    def encode_windows(self, audio: np.ndarray, chunk_samples: int, ctx_samples: int) -> np.ndarray:
        """Batch-encode semantics of `codec_bpe.audio_to_codes --chunk_size_secs --context_secs`
        as the reference scripts drive it (encode_audio_gpu_1.sh:2-8): chunk i is encoded with up
        to ctx_samples of left context ending at the chunk's end, and only the chunk's own
        int(chunk_secs*framerate) codes are kept -- the same rule as
        AudioTokenizer.tokenize_audio (audio_tokenizer.py:72-101)."""
        audio = np.ascontiguousarray(audio, dtype=np.float32)
        Cn, N = audio.shape
        n_chunks = N // chunk_samples
        fpc = int((chunk_samples / self.cfg.sample_rate) * (self.cfg.sample_rate / self.hop))
        out = np.empty((Cn, n_chunks * fpc), np.int64)
        for i in range(n_chunks):
            end = (i + 1) * chunk_samples
            start = max(0, end - max(chunk_samples, ctx_samples))
            codes = self.encode(audio[:, start:end])
            out[:, i * fpc:(i + 1) * fpc] = codes[:, -fpc:]
        return out
