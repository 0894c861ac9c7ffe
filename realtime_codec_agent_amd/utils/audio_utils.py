"""Per-chunk numpy glue of the duplex loop (host side, <= 1920 samples per call).

Same names, argument meaning and results as the reference helpers
(realtime_codec_agent/utils/audio_utils.py:4-46); used by detokenize_output_chunk
(realtime_agent_v2.py:556-579) and run_stream_codes.py:67.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def create_crossfade_ramps(sr: int, fade_secs: float) -> Tuple[int, np.ndarray, np.ndarray]:
    """Equal-power ramps: fade_in[i] = sin(pi/2 * i/L), i in [0, L) (float32); fade_out is its mirror."""
    n = int(sr * fade_secs)
    phase = np.linspace(0, 1, n, endpoint=False, dtype=np.float32)
    fade_in = np.sin(0.5 * np.pi * phase)
    return n, fade_in, fade_in[::-1]


def smooth_join(chunk1: np.ndarray, chunk2: np.ndarray, L: int, fade_in: np.ndarray, fade_out: np.ndarray) -> np.ndarray:
    """Overlap the last L samples of chunk1 with the first L of chunk2 under the ramps; the result
    has len(chunk1) + len(chunk2) - L samples."""
    if chunk1.shape[-1] == 0:
        return chunk2
    if L == 0:
        return np.concatenate((chunk1, chunk2), axis=-1)
    blended = chunk1[..., -L:] * fade_out + chunk2[..., :L] * fade_in
    return np.concatenate((chunk1[..., :-L], blended, chunk2[..., L:]), axis=-1)


def pad_or_trim(chunk: np.ndarray, target_length: int, pad_side: str = "right") -> np.ndarray:
    if chunk.ndim > 1:
        raise ValueError("Input chunk must be a 1D array.")
    n = chunk.shape[-1]
    if n > target_length:
        return chunk[..., :target_length]
    if n < target_length:
        missing = target_length - n
        return np.pad(chunk, (0, missing) if pad_side == "right" else (missing, 0), mode="constant")
    return chunk


def normalize_audio_rms(audio, target_rms=0.05, silence_rms_threshold=0.003):
    rms = np.sqrt(np.mean(audio ** 2))
    if rms < silence_rms_threshold:
        return audio  # treat as silence: leave untouched
    return audio * (target_rms / rms)
