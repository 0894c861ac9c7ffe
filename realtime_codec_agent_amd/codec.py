"""HIP-backed codec model object.

`HipCodec` is the thin handle over the C ABI (include/rca.h, rca_codec_*).
`MagiCodecHIP` wraps it with exactly the attribute surface the reference
AudioTokenizer expects of its `codec_model` (audio_tokenizer.py:26-36,158,189-200;
SURVEY.md 8b-1): codebook_size, sample_rate, pad_audio, encoder,
quantizer.inference, quantizer.codebook.weight, quantizer.codebook_proj, decoder,
eval(), to().  torch is plumbing only: it owns the I/O tensors in HBM and the
stream; every kernel is ours.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np

from . import _native as N
from .codec_model import CodecConfig, init_codec_weights


class HipCodec:
    def __init__(self, cfg: CodecConfig, weights: Dict[str, np.ndarray], device: int = 0):
        self.cfg = cfg
        self.device = device
        self._lib = N.lib()
        self._h = C.c_void_p()
        tensors, keep = N.make_tensors(weights)
        ccfg = N.codec_config_c(cfg)
        N.check(self._lib.rca_codec_create(C.byref(ccfg), tensors, len(weights), device, C.byref(self._h)), "rca_codec_create")
        del keep
        self.hop = cfg.hop

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.rca_codec_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def num_frames(self, T: int) -> int:
        return (T + self.hop - 1) // self.hop

    def set_variant(self, v: int) -> None:
        N.check(self._lib.rca_codec_set_variant(self._h, v), "rca_codec_set_variant")

    def sync(self) -> None:
        N.check(self._lib.rca_codec_sync(self._h), "rca_codec_sync")

    # ---- host-pointer API (numpy in / numpy out)
    def encode(self, pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if pcm.ndim != 2:
            raise ValueError("pcm must be [B,T]")
        B, T = pcm.shape
        codes = np.empty((B, self.num_frames(T)), np.int64)
        N.check(self._lib.rca_codec_encode(self._h, C.c_void_p(pcm.ctypes.data), B, T, C.c_void_p(codes.ctypes.data)), "rca_codec_encode")
        return codes

    def decode(self, codes: np.ndarray) -> np.ndarray:
        codes = np.ascontiguousarray(codes, dtype=np.int64)
        if codes.ndim != 2:
            raise ValueError("codes must be [B,F]")
        B, F = codes.shape
        pcm = np.empty((B, F * self.hop), np.float32)
        N.check(self._lib.rca_codec_decode(self._h, C.c_void_p(codes.ctypes.data), B, F, C.c_void_p(pcm.ctypes.data)), "rca_codec_decode")
        return pcm

    def encode_tail(self, pcm: np.ndarray, n_keep: int) -> np.ndarray:
        """Last n_keep codes of encode(pcm), host buffers, graph-replayed per shape (rca_codec_encode_tail)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if pcm.ndim != 2:
            raise ValueError("pcm must be [B,T]")
        B, T = pcm.shape
        n_keep = min(int(n_keep), self.num_frames(T))
        codes = np.empty((B, n_keep), np.int64)
        N.check(self._lib.rca_codec_encode_tail(self._h, C.c_void_p(pcm.ctypes.data), B, T, n_keep, C.c_void_p(codes.ctypes.data)),
                "rca_codec_encode_tail")
        return codes

    def decode_tail(self, codes: np.ndarray, n_samples: int) -> np.ndarray:
        """Last n_samples of decode(codes), host buffers, graph-replayed per shape (rca_codec_decode_tail)."""
        codes = np.ascontiguousarray(codes, dtype=np.int64)
        if codes.ndim != 2:
            raise ValueError("codes must be [B,F]")
        B, F = codes.shape
        n = min(int(n_samples), F * self.hop)
        pcm = np.empty((B, n), np.float32)
        N.check(self._lib.rca_codec_decode_tail(self._h, C.c_void_p(codes.ctypes.data), B, F, n, C.c_void_p(pcm.ctypes.data)),
                "rca_codec_decode_tail")
        return pcm

    def set_mfma_mode(self, mode: int) -> None:
        """Encoder conv arithmetic: 0 = f32 matrix instruction (bit-exact, default), 3 = bf16 hi + lo split, 1 = bf16 (opt-in, not bit-exact)."""
        N.check(self._lib.rca_codec_set_mfma_mode(self._h, int(mode)), "rca_codec_set_mfma_mode")

    def set_stream_graphs(self, enable: bool) -> None:
        N.check(self._lib.rca_codec_set_stream_graphs(self._h, int(bool(enable))), "rca_codec_set_stream_graphs")

    def codebook(self) -> np.ndarray:
        out = np.empty((self.cfg.codebook_size, self.cfg.codebook_dim), np.float32)
        N.check(self._lib.rca_codec_codebook(self._h, C.c_void_p(out.ctypes.data)), "rca_codec_codebook")
        return out

    def encode_tap(self, pcm: np.ndarray, layer: int) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        B, T = pcm.shape
        F = self.num_frames(T)
        n = self.cfg.n_stages
        L = F * self.hop
        if layer == 0:
            shape = (B, self.cfg.channels[0], L)
        elif layer <= n:
            for i in range(layer):
                L //= self.cfg.strides[i]
            shape = (B, self.cfg.channels[layer], L)
        elif layer == n + 1:
            shape = (B, self.cfg.latent_dim, F)
        else:
            shape = (B * F, self.cfg.codebook_dim)
        out = np.empty(shape, np.float32)
        N.check(self._lib.rca_codec_encode_tap(self._h, C.c_void_p(pcm.ctypes.data), B, T, layer, C.c_void_p(out.ctypes.data),
                                               C.c_int64(out.size)), "rca_codec_encode_tap")
        return out

    # ---- device-pointer API (raw addresses; stream = hipStream_t as int, 0 = handle's own stream)
    def encode_dev(self, pcm_ptr: int, B: int, T: int, codes_ptr: int, stream: int = 0) -> None:
        N.check(self._lib.rca_codec_encode_dev(self._h, C.c_void_p(pcm_ptr), B, T, C.c_void_p(codes_ptr), C.c_void_p(stream)), "rca_codec_encode_dev")

    def decode_dev(self, codes_ptr: int, B: int, F: int, pcm_ptr: int, stream: int = 0) -> None:
        N.check(self._lib.rca_codec_decode_dev(self._h, C.c_void_p(codes_ptr), B, F, C.c_void_p(pcm_ptr), C.c_void_p(stream)), "rca_codec_decode_dev")

    def encode_tail_dev(self, pcm_ptr: int, B: int, T: int, n_keep: int, codes_ptr: int, stream: int = 0) -> None:
        """Last n_keep codes of encode_dev(pcm), computed over their receptive field only (bit-identical)."""
        N.check(self._lib.rca_codec_encode_tail_dev(self._h, C.c_void_p(pcm_ptr), B, T, n_keep, C.c_void_p(codes_ptr), C.c_void_p(stream)),
                "rca_codec_encode_tail_dev")

    def decode_tail_dev(self, codes_ptr: int, B: int, F: int, n_samples: int, pcm_ptr: int, stream: int = 0) -> None:
        """Last n_samples of decode_dev(codes), computed from the codes they depend on only (bit-identical)."""
        N.check(self._lib.rca_codec_decode_tail_dev(self._h, C.c_void_p(codes_ptr), B, F, n_samples, C.c_void_p(pcm_ptr), C.c_void_p(stream)),
                "rca_codec_decode_tail_dev")

    def receptive_field(self) -> Tuple[int, int]:
        """(encoder, decoder) whole frames a kept code / sample can see to its left."""
        a, b = C.c_int32(), C.c_int32()
        N.check(self._lib.rca_codec_receptive_field(self._h, C.byref(a), C.byref(b)), "rca_codec_receptive_field")
        return a.value, b.value

    def set_window_trim(self, enable: bool) -> None:
        """Batch windows encode only what their kept codes can see (same codes, ~10x less work at 2 s context)."""
        N.check(self._lib.rca_codec_set_window_trim(self._h, int(bool(enable))), "rca_codec_set_window_trim")

    def encode_windows_dev(self, audio_ptr: int, Cn: int, Nsamp: int, chunk: int, ctx: int, batch_windows: int, codes_ptr: int,
                           codes_per_channel: int, stream: int = 0) -> None:
        N.check(self._lib.rca_codec_encode_windows_dev(self._h, C.c_void_p(audio_ptr), Cn, C.c_int64(Nsamp), chunk, ctx, batch_windows,
                                                       C.c_void_p(codes_ptr), C.c_int64(codes_per_channel), C.c_void_p(stream)),
                "rca_codec_encode_windows_dev")

    def encode_chunk_range_dev(self, audio_ptr: int, Cn: int, Nsamp: int, chunk: int, ctx: int, batch_windows: int, chunk_begin: int,
                               chunk_end: int, codes_ptr: int, codes_per_channel: int, stream: int = 0) -> None:
        N.check(self._lib.rca_codec_encode_chunk_range_dev(self._h, C.c_void_p(audio_ptr), Cn, C.c_int64(Nsamp), chunk, ctx, batch_windows,
                                                           C.c_int64(chunk_begin), C.c_int64(chunk_end), C.c_void_p(codes_ptr),
                                                           C.c_int64(codes_per_channel), C.c_void_p(stream)),
                "rca_codec_encode_chunk_range_dev")

    def encode_rows_dev(self, audio_ptr: int, src_off_ptr: int, B: int, T: int, n_keep: int, codes_ptr: int, dst_off_ptr: int, span: int,
                        stream: int = 0) -> None:
        """B windows of T samples at audio + src_off[b]; the last n_keep codes of window b land at codes + dst_off[b]."""
        N.check(self._lib.rca_codec_encode_rows_dev(self._h, C.c_void_p(audio_ptr), C.c_void_p(src_off_ptr), B, T, n_keep, C.c_void_p(codes_ptr),
                                                    C.c_void_p(dst_off_ptr), C.c_int64(span), C.c_void_p(stream)), "rca_codec_encode_rows_dev")

    def encoder_dev(self, pcm_ptr: int, B: int, T: int, ze_ptr: int, stream: int = 0) -> None:
        N.check(self._lib.rca_codec_encoder_dev(self._h, C.c_void_p(pcm_ptr), B, T, C.c_void_p(ze_ptr), C.c_void_p(stream)), "rca_codec_encoder_dev")

    def quantize_dev(self, ze_ptr: int, rows: int, codes_ptr: int, stream: int = 0) -> None:
        N.check(self._lib.rca_codec_quantize_dev(self._h, C.c_void_p(ze_ptr), C.c_int64(rows), C.c_void_p(codes_ptr), C.c_void_p(stream)), "rca_codec_quantize_dev")

    def decoder_dev(self, zq_ptr: int, B: int, F: int, pcm_ptr: int, stream: int = 0) -> None:
        N.check(self._lib.rca_codec_decoder_dev(self._h, C.c_void_p(zq_ptr), B, F, C.c_void_p(pcm_ptr), C.c_void_p(stream)), "rca_codec_decoder_dev")

    def codebook_dev_ptr(self) -> int:
        p = C.c_void_p()
        N.check(self._lib.rca_codec_codebook_dev(self._h, C.byref(p)), "rca_codec_codebook_dev")
        return p.value

    def profile(self, enable: bool) -> None:
        N.check(self._lib.rca_codec_profile(self._h, 1 if enable else 0), "rca_codec_profile")

    def profile_read(self, kclass: int) -> dict:
        ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
        N.check(self._lib.rca_codec_profile_read(self._h, kclass, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)), "rca_codec_profile_read")
        return dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)

    def frames_per_chunk(self, chunk_samples: int) -> int:
        # int(audio_secs * framerate) as audio_tokenizer.py:99-100
        return int((chunk_samples / self.cfg.sample_rate) * self.cfg.framerate)


# ----------------------------------------------------------------------------------------------
def _torch():
    import torch
    return torch


def _stream_of(t) -> int:
    torch = _torch()
    return int(torch.cuda.current_stream(t.device).cuda_stream)


class _Codebook:
    def __init__(self, weight):
        self.weight = weight


class _Quantizer:
    """quantizer.inference / .codebook.weight / .codebook_proj (audio_tokenizer.py:158,192,198)."""

    def __init__(self, model: "MagiCodecHIP"):
        self._m = model
        torch = _torch()
        self.codebook = _Codebook(torch.from_numpy(model._weights["quantizer.codebook.weight"]).to(model.device))
        self._cb_proj = None

    def projected_codebook(self):
        if self._cb_proj is None:
            torch = _torch()
            # the handle computed codebook_proj(codebook.weight) once at load; fetch that constant
            self._cb_proj = torch.from_numpy(self._m.hip.codebook()).to(self._m.device)
        return self._cb_proj

    def codebook_proj(self, w):
        # The projection of the model's own codebook is a constant computed once at load
        # (rca_codec_create); any other input goes through the same linear map on the device.
        if w is self.codebook.weight:
            return self.projected_codebook()
        torch = _torch()
        pw = torch.from_numpy(self._m._weights["quantizer.codebook_proj.weight"]).to(w.device)
        pb = torch.from_numpy(self._m._weights["quantizer.codebook_proj.bias"]).to(w.device)
        return torch.nn.functional.linear(w.float(), pw, pb)

    def inference(self, z_e):
        torch = _torch()
        B, F, D = z_e.shape
        z_e = z_e.contiguous().float()
        idx = torch.empty((B, F), dtype=torch.int64, device=z_e.device)
        self._m.hip.quantize_dev(z_e.data_ptr(), B * F, idx.data_ptr(), _stream_of(z_e))
        z_q = torch.nn.functional.embedding(idx, self.projected_codebook())
        return z_q, idx


class MagiCodecHIP:
    """Model object with the MagiCodec attribute surface, every op a HIP kernel of this repo."""

    def __init__(self, cfg: Optional[CodecConfig] = None, weights: Optional[Dict[str, np.ndarray]] = None, device=None, seed: int = 0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise N.RcaError("MagiCodecHIP needs a GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.cfg = cfg or CodecConfig()
        self._weights = weights if weights is not None else init_codec_weights(self.cfg, seed)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        elif not isinstance(device, torch.device):
            device = torch.device(device)
        if device.type != "cuda":
            raise N.RcaError(f"MagiCodecHIP cannot live on device '{device}'")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.hip = HipCodec(self.cfg, self._weights, device.index)
        self.codebook_size = self.cfg.codebook_size
        self.sample_rate = self.cfg.sample_rate
        self.hop = self.cfg.hop
        self.quantizer = _Quantizer(self)

    # nn.Module-like no-ops used by the reference (audio_tokenizer.py:28)
    def eval(self):
        return self

    def to(self, device):
        torch = _torch()
        d = torch.device(device) if not isinstance(device, torch.device) else device
        if d.type != "cuda" or (d.index is not None and d.index != self.device.index):
            raise N.RcaError(f"MagiCodecHIP is bound to {self.device}; cannot move to {d}")
        return self

    def pad_audio(self, x):
        torch = _torch()
        pad = (-x.shape[-1]) % self.hop
        return torch.nn.functional.pad(x, (0, pad)) if pad else x

    def encoder(self, x):
        torch = _torch()
        x = x.contiguous().float()
        B, T = x.shape
        F = self.hip.num_frames(T)
        ze = torch.empty((B, F, self.cfg.latent_dim), dtype=torch.float32, device=x.device)
        self.hip.encoder_dev(x.data_ptr(), B, T, ze.data_ptr(), _stream_of(x))
        return ze

    def decoder(self, z_q):
        torch = _torch()
        z_q = z_q.contiguous().float()
        B, F, J = z_q.shape
        pcm = torch.empty((B, 1, F * self.hop), dtype=torch.float32, device=z_q.device)
        self.hip.decoder_dev(z_q.data_ptr(), B, F, pcm.data_ptr(), _stream_of(z_q))
        return pcm

    # fused fast paths used by this repo's AudioTokenizer
    def encode_codes(self, x):
        """pad_audio -> encoder -> quantizer.inference in one C-ABI call: [B,T] f32 -> [B,F] int64."""
        torch = _torch()
        x = x.contiguous().float()
        B, T = x.shape
        codes = torch.empty((B, self.hip.num_frames(T)), dtype=torch.int64, device=x.device)
        self.hip.encode_dev(x.data_ptr(), B, T, codes.data_ptr(), _stream_of(x))
        return codes

    def encode_tail(self, x, n_keep: int):
        """The last n_keep columns of encode_codes(x), bit-identical, at the cost of n_keep + margin frames:
        what tokenize_audio keeps of each rolling window (audio_tokenizer.py:98-101).  [B,T] f32 -> [B,n_keep] int64."""
        torch = _torch()
        x = x.contiguous().float()
        B, T = x.shape
        n_keep = min(int(n_keep), self.hip.num_frames(T))
        codes = torch.empty((B, n_keep), dtype=torch.int64, device=x.device)
        self.hip.encode_tail_dev(x.data_ptr(), B, T, n_keep, codes.data_ptr(), _stream_of(x))
        return codes

    def encode_tail_np(self, x: np.ndarray, n_keep: int) -> np.ndarray:
        """encode_tail on host arrays, no torch in the loop: what AudioTokenizer calls once per frame."""
        return self.hip.encode_tail(x, n_keep)

    def decode_tail_np(self, codes: np.ndarray, n_samples: int) -> np.ndarray:
        """decode_tail on host arrays: [B,F] int64 -> [B,min(n_samples, F*hop)] f32."""
        return self.hip.decode_tail(codes, n_samples)

    def decode_tail(self, codes, n_samples: int):
        """The last n_samples of decode_codes(codes), bit-identical (audio_tokenizer.py:141-145).
        [B,F] int64 -> [B,1,min(n_samples, F*hop)] f32."""
        torch = _torch()
        codes = codes.contiguous().to(torch.int64)
        B, F = codes.shape
        n = min(int(n_samples), F * self.hop)
        pcm = torch.empty((B, 1, n), dtype=torch.float32, device=codes.device)
        self.hip.decode_tail_dev(codes.data_ptr(), B, F, n, pcm.data_ptr(), _stream_of(codes))
        return pcm

    def decode_codes(self, codes):
        """embedding(codes, projected codebook) -> decoder in one C-ABI call: [B,F] int64 -> [B,1,T] f32."""
        torch = _torch()
        codes = codes.contiguous().to(torch.int64)
        B, F = codes.shape
        pcm = torch.empty((B, 1, F * self.hop), dtype=torch.float32, device=codes.device)
        self.hip.decode_dev(codes.data_ptr(), B, F, pcm.data_ptr(), _stream_of(codes))
        return pcm


_MODEL_REGISTRY = {"MagiCodec-50Hz-Base": CodecConfig}


def load_magicodec_model(name: str, device, seed: int = 0) -> Tuple[MagiCodecHIP, None, None]:
    """Stand-in for codec_bpe.tools.codec_utils.load_magicodec_model (audio_tokenizer.py:8,27):
    same (model, _, _) return shape.  No checkpoint exists offline, so the named architecture
    is built with seeded random weights."""
    if name not in _MODEL_REGISTRY:
        raise ValueError(f"unknown codec model '{name}' (known: {sorted(_MODEL_REGISTRY)})")
    return MagiCodecHIP(_MODEL_REGISTRY[name](), device=device, seed=seed), None, None
