"""AudioTokenizer -- the streaming face of the codec: PCM chunks in, code characters out, and back.

Public surface and rolling-context rules of the reference class (realtime_codec_agent/audio_tokenizer.py:10-215;
SURVEY.md 8b-2): same constructor, methods, attributes and edge cases, so `run_stream_codes.py`, `run_demo.py` and the
agent drive it unchanged.  Inside, it is organised around the two rolling windows the rules describe:

  PcmWindow   the last max(chunk, context_secs) of input audio per channel.  tokenize_audio appends a chunk, the codec
              sees the window, and only the code frames that belong to the new chunk are returned (:67-103).
  CodeWindow  the last max(chunk, context_frames) code characters, channel-interleaved.  detokenize_audio appends a
              chunk of characters, the codec decodes the window, and only the samples of the new chunk (plus the
              caller's preroll) are returned (:105-149).

How the codec is driven:
  * `codec_model` given by NAME builds the HIP model (MagiCodecHIP) and fails loudly without a GPU / the built
    library; an already-constructed model OBJECT is used as is (the reference allows both, :26-28, and
    clone_for_self_play relies on it, realtime_agent_resources.py:46).
  * streaming shortcut: a model offering encode_tail_np / decode_tail_np is asked for exactly the
    kept frames / samples -- bit-identical output, computed over their receptive field instead of the whole window,
    host arrays in and out, one captured graph per call shape.  `streaming_tail = False` switches it off.
  * otherwise a model with encode_codes / decode_codes is driven through those fused calls, and any other object
    through pad_audio -> encoder -> quantizer.inference and embedding -> decoder, the reference's own sequence
    (:189-201).
  * arithmetic is fp32 on the device (the reference autocasts to bf16 on CUDA, :24,78-82).
"""
from __future__ import annotations

import math
from typing import Any, List, Optional, Tuple, Union

import numpy as np
import torch

from .codec_chars import UNICODE_OFFSET_LARGE, chars_to_codes, codes_to_chars

AudioArg = Union[Tuple[int, np.ndarray], np.ndarray]


def _to_mono(audio: np.ndarray) -> np.ndarray:
    """librosa.to_mono: mean over the leading (channel) axis."""
    return audio.mean(axis=0) if audio.ndim > 1 else audio


def _resample(audio: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    """The reference calls librosa.resample (soxr_hq), which is not installed here; a polyphase resampler stands in.
    Off the hot path: the agent always feeds audio at the codec rate."""
    from scipy.signal import resample_poly
    g = math.gcd(int(orig_sr), int(target_sr))
    return resample_poly(audio, int(target_sr) // g, int(orig_sr) // g, axis=-1).astype(np.float32)


class PcmWindow:
    """[channels, samples] float32, keeps the newest max(len(chunk), limit) samples."""

    def __init__(self, channels: int, limit: int):
        self.limit = limit
        self.data = np.zeros((channels, 0), dtype=np.float32)

    def push(self, chunk: np.ndarray) -> np.ndarray:
        chunk = chunk.reshape(self.data.shape[0], -1)
        keep = max(chunk.shape[-1], self.limit)
        self.data = np.concatenate((self.data, chunk), axis=-1)[..., -keep:]
        return self.data


class CodeWindow:
    """Channel-interleaved code characters, keeps the newest max(len(chunk), limit) of them."""

    def __init__(self, channels: int, limit: int):
        self.channels = channels
        self.limit = limit
        self.text = ""

    def push(self, chars: str) -> str:
        keep = max(len(chars), self.limit)
        self.text = (self.text + chars)[-keep:]
        return self.text

    def per_channel(self) -> List[str]:
        return [self.text[c::self.channels] for c in range(self.channels)]


class AudioTokenizer:
    def __init__(
        self,
        codec_model: Union[str, Any] = "MagiCodec-50Hz-Base",
        num_channels: int = 1,
        context_secs: float = 2.0,
        unicode_offset: int = UNICODE_OFFSET_LARGE,
        device: Optional[Union[str, torch.device]] = None,
    ):
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device) if isinstance(device, str) else device
        self.autocast_bfloat16 = False  # fp32 kernels; see module docstring

        if isinstance(codec_model, str):
            from .codec import load_magicodec_model
            codec_model, _, _ = load_magicodec_model(codec_model, self.device)
        self.codec_model = codec_model.eval().to(self.device)
        self.streaming_tail = True   # attribute, not a ctor argument: the constructor keeps the reference signature

        self.num_channels = num_channels
        self.num_codebooks = 1
        self.codebook_size = self.codec_model.codebook_size
        self.context_secs = context_secs
        self.unicode_offset = unicode_offset
        self.sampling_rate = self.codec_model.sample_rate
        self.framerate = self._compute_framerate()
        self.context_samples = int(self.context_secs * self.sampling_rate)
        self.context_frames = int(self.context_secs * self.framerate * self.num_channels)
        self.reset_context()

    # ------------------------------------------------------------------ rolling state
    def reset_context(self):
        self._pcm = PcmWindow(self.num_channels, self.context_samples)
        self._codes = CodeWindow(self.num_channels, self.context_frames)

    @property
    def tokenize_context(self) -> np.ndarray:
        return self._pcm.data

    @tokenize_context.setter
    def tokenize_context(self, value: np.ndarray) -> None:
        self._pcm.data = np.asarray(value, dtype=np.float32)

    @property
    def detokenize_context(self) -> str:
        return self._codes.text

    @detokenize_context.setter
    def detokenize_context(self, value: str) -> None:
        self._codes.text = value

    def get_audio_codes_str_secs(self, audio_codes_str: str) -> float:
        return len(audio_codes_str) / (self.framerate * self.num_channels)

    # ------------------------------------------------------------------ PCM -> characters
    def chunked_tokenize_audio(self, audio: AudioArg, chunk_size_secs: float) -> str:
        sr, pcm = (self.sampling_rate, audio) if isinstance(audio, np.ndarray) else audio
        step = int(chunk_size_secs * sr)
        return "".join(self.tokenize_audio((sr, pcm[..., i:i + step])) for i in range(0, pcm.shape[-1], step))

    @torch.inference_mode()
    def tokenize_audio(self, audio: AudioArg) -> str:
        chunk = self._prep_audio_for_tokenization(audio)
        window = self._pcm.push(chunk)
        # characters owed for this chunk (reference :98-100); note [-0:] keeps everything, as the reference does
        n_chars = int(chunk.shape[-1] / self.sampling_rate * self.framerate * self.num_channels)
        model = self.codec_model
        if self.streaming_tail and n_chars > 0 and hasattr(model, "encode_tail_np"):
            frames = -(-n_chars // self.num_channels)
            codes = model.encode_tail_np(np.ascontiguousarray(window), frames)[:, None, :]          # [C, 1, frames]
        else:
            # every channel is one row of a single batched launch (the reference loops channels, :84)
            codes = self._magicodec_encode(torch.from_numpy(np.ascontiguousarray(window)).to(self.device)).cpu().numpy()
        per_channel = [codes_to_chars(c, self.codebook_size, unicode_offset=self.unicode_offset) for c in codes]
        interleaved = "".join(ch for frame in zip(*per_channel) for ch in frame)
        return interleaved[-n_chars:]

    # ------------------------------------------------------------------ characters -> PCM
    @torch.inference_mode()
    def detokenize_audio(self, audio_codes_str: str, preroll_samples: int = 0) -> Tuple[Tuple[int, np.ndarray], str, int]:
        audio_codes_str, end_hanging = self._drop_hanging_channel_codes(audio_codes_str)
        self._codes.push(audio_codes_str)
        rows = self._codes.per_channel()
        # samples owed for this chunk plus the caller's preroll (reference :141-145)
        n_samples = int(self.get_audio_codes_str_secs(audio_codes_str) * self.sampling_rate) + preroll_samples
        model = self.codec_model
        if self.streaming_tail and n_samples > 0 and rows[0] and hasattr(model, "decode_tail_np"):
            ids = np.stack([chars_to_codes(r, self.num_codebooks, self.codebook_size, unicode_offset=self.unicode_offset)[0] for r in rows])
            pcm = model.decode_tail_np(ids, n_samples)[None]                                        # [1, C, n]
        else:
            ids = torch.stack([chars_to_codes(r, self.num_codebooks, self.codebook_size, return_tensors="pt",
                                              unicode_offset=self.unicode_offset) for r in rows]).to(self.device)   # [C, 1, F]
            # [1, C, T] == cat(dim=1) of the per-channel [1, 1, T] (reference :136-139)
            pcm = self._magicodec_decode(ids).transpose(0, 1).cpu().numpy()
        pcm = pcm[..., -n_samples:]
        preroll_left = max(0, preroll_samples - n_samples + pcm.shape[-1])
        return (self.sampling_rate, pcm[0, 0] if self.num_channels == 1 else pcm[0]), end_hanging, preroll_left

    # ------------------------------------------------------------------ one-replay duplex frame (rca_duplex_frame)
    def duplex_plan(self, audio: AudioArg, preroll_samples: int = 0) -> Optional[dict]:
        """What tokenize_audio(audio) followed by detokenize_audio(<as many codes>, preroll_samples) would hand to the codec, WITHOUT
        touching the rolling state: the PCM window after the push, the code context the new codes will be appended to, the samples
        owed.  None unless both windows are in steady state (full: one call shape, one captured graph), the stream is mono and the
        codec object is the HIP one -- the caller then makes the two separate calls.  duplex_commit_pcm / duplex_commit_codes apply
        the state changes of the two calls once the fused frame has run."""
        model = self.codec_model
        if not (self.streaming_tail and self.num_channels == 1 and hasattr(model, "hip")):
            return None
        chunk = self._prep_audio_for_tokenization(audio).reshape(1, -1)
        n_new = chunk.shape[-1]
        n_codes = int(n_new / self.sampling_rate * self.framerate * self.num_channels)
        if n_codes < 1 or n_codes > 8:
            return None
        window = np.concatenate((self._pcm.data, chunk), axis=-1)[..., -max(n_new, self._pcm.limit):]
        have = len(self._codes.text)
        F = min(have + n_codes, max(n_codes, self._codes.limit))
        n_samples = int(n_codes / (self.framerate * self.num_channels) * self.sampling_rate) + preroll_samples
        if window.shape[-1] != self.context_samples or F != self.context_frames or n_samples > F * model.hip.hop:
            return None
        ctx = self._codes.text[have + n_codes - F:]
        code_ctx = chars_to_codes(ctx, self.num_codebooks, self.codebook_size, unicode_offset=self.unicode_offset)[0]
        return {"window": np.ascontiguousarray(window), "n_codes": n_codes, "code_ctx": np.ascontiguousarray(code_ctx, dtype=np.int64),
                "n_samples": n_samples, "preroll": preroll_samples}

    def duplex_commit_pcm(self, plan: dict) -> None:
        """tokenize_audio's state change: the pushed window."""
        self._pcm.data = plan["window"]

    def duplex_commit_codes(self, plan: dict, audio_codes_str: str, pcm: np.ndarray) -> Tuple[Tuple[int, np.ndarray], str, int]:
        """detokenize_audio's state change and return value, given the decode tail the fused frame produced."""
        self._codes.push(audio_codes_str)
        return (self.sampling_rate, pcm), "", plan["preroll"]

    @torch.inference_mode()
    def get_codec_embeddings(self) -> torch.Tensor:
        q = self.codec_model.quantizer
        return q.codebook_proj(q.codebook.weight)

    def _drop_hanging_channel_codes(self, audio_str: str) -> Tuple[str, str]:
        """Cut a trailing partial frame.  Reproduces the reference as written (:161-168): `end_hanging` is sliced AFTER the
        truncation, so it is the tail of the KEPT string, not the characters that were dropped."""
        extra = len(audio_str) % self.num_channels
        if extra == 0:
            return audio_str, ""
        kept = audio_str[:-extra]
        return kept, kept[-extra:]

    # ------------------------------------------------------------------ codec plumbing
    @torch.inference_mode()
    def _encode_silence(self, secs: float) -> torch.Tensor:
        return self._magicodec_encode(torch.zeros(1, int(secs * self.sampling_rate), device=self.device))

    def _compute_framerate(self) -> float:
        """sr / ceil(samples / frames) on 10 s of silence, like the reference (:181-187)."""
        secs = 10.0
        frames = self._encode_silence(secs).shape[-1]
        return self.sampling_rate / math.ceil(int(secs * self.sampling_rate) / frames)

    def _magicodec_encode(self, x: torch.Tensor) -> torch.Tensor:
        """x [B,T] -> indices [B,1,F] (codebook dimension added, reference :189-194)."""
        m = self.codec_model
        if hasattr(m, "encode_codes"):
            return m.encode_codes(x).unsqueeze(1)
        _, idx = m.quantizer.inference(m.encoder(m.pad_audio(x)))
        return idx.unsqueeze(1)

    def _magicodec_decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [B,1,F] -> recon [B,1,T] float32 (reference :196-201)."""
        m = self.codec_model
        codes = codes.squeeze(1)
        if hasattr(m, "decode_codes"):
            return m.decode_codes(codes).float()
        table = m.quantizer.codebook_proj(m.quantizer.codebook.weight)
        return m.decoder(torch.nn.functional.embedding(codes, table)).float()

    def _prep_audio_for_tokenization(self, audio: AudioArg) -> np.ndarray:
        sr, pcm = (self.sampling_rate, audio) if isinstance(audio, np.ndarray) else audio
        if pcm.dtype == np.int16:
            pcm = pcm.astype("float32") / 32768.0
        if self.num_channels == 1 and pcm.ndim > 1:
            pcm = _to_mono(pcm)
        if sr != self.sampling_rate:
            pcm = _resample(pcm, orig_sr=sr, target_sr=self.sampling_rate)
        return pcm.astype(np.float32, copy=False)
