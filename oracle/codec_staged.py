"""The codec as the MODEL OBJECT the reference AudioTokenizer drives (TEST INFRASTRUCTURE ONLY).

reference audio_tokenizer.py touches exactly these members of `codec_model` (:26-36,158,189-200;
SURVEY.md 8b-1): eval(), to(device), codebook_size, sample_rate, pad_audio, encoder,
quantizer.inference, quantizer.codebook.weight, quantizer.codebook_proj, decoder.  Here every
one of them runs on the C oracle's primitives (oracle/codec_oracle.c), so a reference
AudioTokenizer built over this object (tests/golden/make_tokenizer_golden.py) emits the oracle's
code ids and PCM, which the HIP path reproduces bit for bit.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .codec import OracleCodec, _fp, lib


class _Codebook:
    def __init__(self, raw: np.ndarray):
        self.weight = torch.from_numpy(raw)


class _Quantizer:
    def __init__(self, oc: OracleCodec):
        self._oc = oc
        self.codebook = _Codebook(oc._w["quantizer.codebook.weight"])

    def codebook_proj(self, weight: torch.Tensor) -> torch.Tensor:
        """Linear(raw -> cd) over the given rows (the reference passes codebook.weight, :158,198)."""
        oc, cfg = self._oc, self._oc.cfg
        raw = np.ascontiguousarray(weight.detach().cpu().numpy(), dtype=np.float32)
        out = np.empty((raw.shape[0], cfg.codebook_dim), np.float32)
        lib().oracle_linear(_fp(raw), C.c_long(raw.shape[0]), cfg.codebook_raw_dim, _fp(oc._w["quantizer.codebook_proj.weight"]),
                            _fp(oc._w["quantizer.codebook_proj.bias"]), cfg.codebook_dim, _fp(out))
        return torch.from_numpy(out)

    def inference(self, z_e: torch.Tensor):
        """z_e [B,F,D] -> (z_q [B,F,cd], idx [B,F] int64): in_proj, then the nearest projected code."""
        oc, cfg = self._oc, self._oc.cfg
        ze = np.ascontiguousarray(z_e.detach().cpu().numpy(), dtype=np.float32)
        B, F, D = ze.shape
        z = np.empty((B * F, cfg.codebook_dim), np.float32)
        lib().oracle_linear(_fp(ze), C.c_long(B * F), D, _fp(oc._w["quantizer.in_proj.weight"]), _fp(oc._w["quantizer.in_proj.bias"]),
                            cfg.codebook_dim, _fp(z))
        idx = np.empty((B * F,), np.int64)
        lib().oracle_vq_argmax(_fp(z), C.c_long(B * F), _fp(oc.cb), _fp(oc.hc), cfg.codebook_size, cfg.codebook_dim,
                               idx.ctypes.data_as(C.POINTER(C.c_int64)))
        idx = idx.reshape(B, F)
        return torch.from_numpy(oc.cb[idx]), torch.from_numpy(idx)


class OracleStagedCodecModel:
    def __init__(self, oc: OracleCodec):
        self.oc = oc
        self.codebook_size = oc.cfg.codebook_size
        self.sample_rate = oc.cfg.sample_rate
        self.quantizer = _Quantizer(oc)
        self.calls = dict(pad_audio=0, encoder=0, inference=0, decoder=0)

    def eval(self):
        return self

    def to(self, device):
        return self

    def pad_audio(self, x: torch.Tensor) -> torch.Tensor:
        self.calls["pad_audio"] += 1
        pad = (-x.shape[-1]) % self.oc.hop
        return torch.nn.functional.pad(x, (0, pad)) if pad else x

    def encoder(self, x: torch.Tensor) -> torch.Tensor:
        """padded PCM [B,T] -> z_e [B,F,D] (the conv_out activation of the oracle's encoder stack)."""
        self.calls["encoder"] += 1
        pcm = np.ascontiguousarray(x.detach().cpu().numpy(), dtype=np.float32)
        assert pcm.shape[-1] % self.oc.hop == 0
        _, ze = self.oc.encode(pcm, tap_layer=len(self.oc.cfg.strides) + 1)      # [B, D, F]
        return torch.from_numpy(np.ascontiguousarray(ze.transpose(0, 2, 1)))

    def decoder(self, z_q: torch.Tensor) -> torch.Tensor:
        """z_q [B,F,cd] -> [B,1,T]"""
        self.calls["decoder"] += 1
        zq = np.ascontiguousarray(z_q.detach().cpu().numpy(), dtype=np.float32)
        B, F, _ = zq.shape
        pcm = np.empty((B, F * self.oc.hop), np.float32)
        lib().oracle_codec_decoder(C.byref(self.oc._c), C.byref(self.oc._wstruct), _fp(zq), B, F, _fp(pcm))
        return torch.from_numpy(pcm).unsqueeze(1)
