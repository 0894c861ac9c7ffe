"""AudioTokenizer -- streaming wrapper around the codec, same public surface and rolling-context
semantics as the reference class (realtime_codec_agent/audio_tokenizer.py:10-215).

Differences, all behind the same interface:
  * `codec_model` given by NAME builds the HIP model (MagiCodecHIP) and fails loudly without a
    GPU / the built library; an already-constructed model OBJECT is used as is, exactly as the
    reference allows (audio_tokenizer.py:26-28; clone_for_self_play, realtime_agent_resources.py:46).
  * a model exposing encode_codes()/decode_codes() (ours) is driven through those fused C-ABI
    calls; any other object is driven through pad_audio/encoder/quantizer.inference/decoder, the
    reference's own sequence (audio_tokenizer.py:189-201).
  * arithmetic is fp32 on the device (the reference autocasts to bf16 on CUDA, :24,78-82).
"""
from __future__ import annotations

import itertools
import math
from typing import Any, Optional, Tuple, Union

import numpy as np
import torch

from .codec_chars import UNICODE_OFFSET_LARGE, chars_to_codes, codes_to_chars


def _to_mono(audio: np.ndarray) -> np.ndarray:
    # librosa.to_mono: mean over the leading (channel) axis
    return audio.mean(axis=0) if audio.ndim > 1 else audio


def _resample(audio: np.ndarray, orig_sr: int, target_sr: int) -> np.ndarray:
    # The reference calls librosa.resample (soxr_hq), which is not installed here; a polyphase
    # resampler stands in.  Off the hot path: the agent always feeds audio at the codec rate.
    from scipy.signal import resample_poly
    g = math.gcd(int(orig_sr), int(target_sr))
    return resample_poly(audio, int(target_sr) // g, int(orig_sr) // g, axis=-1).astype(np.float32)


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
            device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        elif isinstance(device, str):
            device = torch.device(device)
        self.device = device
        self.autocast_bfloat16 = False  # fp32 kernels; see module docstring

        if isinstance(codec_model, str):
            from .codec import load_magicodec_model
            codec_model, _, _ = load_magicodec_model(codec_model, self.device)
        self.codec_model = codec_model.eval().to(self.device)

        # Streaming shortcut: when the codec model offers encode_tail / decode_tail, each call computes
        # only the frames / samples that are kept (their receptive field), not the whole rolling window.  Output is
        # bit-identical; set to False to run every window in full like the reference (:74,:113).
        self.streaming_tail = True

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

    def reset_context(self):
        self.tokenize_context = np.zeros((self.num_channels, 0), dtype=np.float32)
        self.detokenize_context = ""

    def get_audio_codes_str_secs(self, audio_codes_str: str) -> float:
        secs = len(audio_codes_str) / (self.framerate * self.num_channels)
        return secs

    def chunked_tokenize_audio(self, audio: Union[Tuple[int, np.ndarray], np.ndarray], chunk_size_secs: float) -> str:
        if isinstance(audio, np.ndarray):
            sr = self.sampling_rate
        else:
            sr, audio = audio
        chunk_size_samples = int(chunk_size_secs * sr)
        chunk_codes_strs = []
        for start in range(0, audio.shape[-1], chunk_size_samples):
            end = start + chunk_size_samples
            chunk = audio[..., start:end]
            chunk_codes_strs.append(self.tokenize_audio((sr, chunk)))
        return "".join(chunk_codes_strs)

    @torch.inference_mode()
    def tokenize_audio(self, audio: Union[Tuple[int, np.ndarray], np.ndarray]) -> str:
        audio = self._prep_audio_for_tokenization(audio)

        # append audio to the context, trim to max(context, audio) samples (reference :72-74)
        self.tokenize_context = np.concatenate((self.tokenize_context, audio.reshape(self.num_channels, -1)), axis=-1)
        self.tokenize_context = self.tokenize_context[..., -max(audio.shape[-1], self.context_samples):]

        audio_secs = audio.shape[-1] / self.sampling_rate
        audio_frames = int(audio_secs * self.framerate * self.num_channels)
        # every channel is one row of a single batched launch (the reference loops channels, :84)
        if self.streaming_tail and audio_frames > 0 and hasattr(self.codec_model, "encode_tail_np"):
            # only the last audio_frames characters survive below: ask the codec for exactly those frames.  Same
            # codes, bit for bit; the encoder runs over their receptive field instead of the whole window, and the
            # call (copies included) replays as one captured graph.
            n_keep = -(-audio_frames // self.num_channels)
            encoder_outputs = self.codec_model.encode_tail_np(np.ascontiguousarray(self.tokenize_context), n_keep)[:, None, :]
        else:
            input_audio = torch.from_numpy(np.ascontiguousarray(self.tokenize_context)).to(self.device)
            encoder_outputs = self._magicodec_encode(input_audio).cpu().numpy()  # [C, 1, F]

        channels_chars = [codes_to_chars(ch_codes, self.codebook_size, unicode_offset=self.unicode_offset) for ch_codes in encoder_outputs]
        audio_codes_str = "".join(list(itertools.chain.from_iterable(zip(*channels_chars))))

        # discard context codes that come before the audio we are tokenizing (reference :98-101)
        audio_codes_str = audio_codes_str[-audio_frames:]
        return audio_codes_str

    @torch.inference_mode()
    def detokenize_audio(self, audio_codes_str: str, preroll_samples: int = 0) -> Tuple[Tuple[int, np.ndarray], str, int]:
        audio_codes_str, end_hanging = self._drop_hanging_channel_codes(audio_codes_str)

        self.detokenize_context += audio_codes_str
        self.detokenize_context = self.detokenize_context[-max(len(audio_codes_str), self.context_frames):]

        input_audio_codes_str = [self.detokenize_context[i::self.num_channels] for i in range(self.num_channels)]

        # discard context audio that comes before the codes we are detokenizing (reference :141-145)
        audio_secs = self.get_audio_codes_str_secs(audio_codes_str)
        audio_samples = int(audio_secs * self.sampling_rate) + preroll_samples
        if self.streaming_tail and audio_samples > 0 and len(input_audio_codes_str[0]) > 0 and hasattr(self.codec_model, "decode_tail_np"):
            # the same samples, bit for bit, decoded from the codes they depend on instead of the whole context
            codes_np = np.stack([
                chars_to_codes(ch_chars, self.num_codebooks, self.codebook_size, unicode_offset=self.unicode_offset)[0]
                for ch_chars in input_audio_codes_str
            ])  # [C, F]
            output_audio = self.codec_model.decode_tail_np(codes_np, audio_samples)[None]  # [1, C, n]
        else:
            input_audio_codes = [
                chars_to_codes(ch_chars, self.num_codebooks, self.codebook_size, return_tensors="pt", unicode_offset=self.unicode_offset)
                for ch_chars in input_audio_codes_str
            ]
            input_audio_codes = torch.stack(input_audio_codes).to(self.device)  # [C, 1, F]
            output_audio = self._magicodec_decode(input_audio_codes)  # [C, 1, T]
            # [1, C, T] == cat(dim=1) of the per-channel [1,1,T] (reference :136-139)
            output_audio = output_audio.transpose(0, 1).cpu().numpy()
        output_audio = output_audio[..., -audio_samples:]
        preroll_samples = max(0, preroll_samples - audio_samples + output_audio.shape[-1])

        output_audio = output_audio[0, 0] if self.num_channels == 1 else output_audio[0]
        return (self.sampling_rate, output_audio), end_hanging, preroll_samples

    @torch.inference_mode()
    def get_codec_embeddings(self) -> torch.Tensor:
        return self.codec_model.quantizer.codebook_proj(self.codec_model.quantizer.codebook.weight)

    def _drop_hanging_channel_codes(self, audio_str: str) -> Tuple[str, str]:
        div_rem = len(audio_str) % self.num_channels
        if div_rem != 0:
            # NOTE: reproduces the reference as written (:161-168): end_hanging is sliced AFTER the
            # truncation, so it is the tail of the KEPT string, not the dropped characters.
            audio_str = audio_str[:-div_rem]
            end_hanging = audio_str[-div_rem:]
        else:
            end_hanging = ""
        return audio_str, end_hanging

    @torch.inference_mode()
    def _encode_silence(self, secs: float) -> torch.Tensor:
        audio = torch.zeros(int(secs * self.sampling_rate)).to(self.device)
        return self._magicodec_encode(audio.unsqueeze(0))

    def _compute_framerate(self) -> float:
        test_secs = 10.0
        audio_codes = self._encode_silence(test_secs)
        samples = int(test_secs * self.sampling_rate)
        samples_per_frame = math.ceil(samples / audio_codes.shape[-1])
        return self.sampling_rate / samples_per_frame

    def _magicodec_encode(self, x: torch.Tensor) -> torch.Tensor:
        """x [B,T] -> indices [B,1,F] (codebook dimension added, reference :189-194)."""
        m = self.codec_model
        if hasattr(m, "encode_codes"):
            quantized_indices = m.encode_codes(x)
        else:
            x = m.pad_audio(x)
            z_e = m.encoder(x)
            _, quantized_indices = m.quantizer.inference(z_e)
        return quantized_indices.unsqueeze(1)

    def _magicodec_decode(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [B,1,F] -> recon [B,1,T] float32 (reference :196-201)."""
        m = self.codec_model
        codes = codes.squeeze(1)
        if hasattr(m, "decode_codes"):
            return m.decode_codes(codes).float()
        codebook = m.quantizer.codebook_proj(m.quantizer.codebook.weight)
        z_q = torch.nn.functional.embedding(codes, codebook)
        return m.decoder(z_q).float()

    def _prep_audio_for_tokenization(self, audio: Union[Tuple[int, np.ndarray], np.ndarray]) -> np.ndarray:
        if isinstance(audio, np.ndarray):
            orig_sr = self.sampling_rate
        else:
            orig_sr, audio = audio
        if audio.dtype == np.int16:
            audio = audio.astype("float32") / 32768.0
        if self.num_channels == 1 and audio.ndim > 1:
            audio = _to_mono(audio)
        if orig_sr != self.sampling_rate:
            audio = _resample(audio, orig_sr=orig_sr, target_sr=self.sampling_rate)
        return audio.astype(np.float32, copy=False)
