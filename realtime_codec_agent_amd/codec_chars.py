"""codes <-> unicode chars, the wire format between the codec and the LM tokenizer.

Restates the two helpers the reference imports from the third-party `codec_bpe` package
(audio_tokenizer.py:7; call sites :89-95 and :119-127): every code of codebook i becomes the
single character chr(unicode_offset + i*codebook_size + code), codebooks interleaved per
frame.  UNICODE_OFFSET_LARGE = 0xE000 is inferred from the dataset scripts, which pass
--unicode_offset=0xE000 for the same 131072-entry vocabulary (prep_lm_dataset_magicodec.sh:4).
"""
from __future__ import annotations

from typing import Union

import numpy as np

UNICODE_OFFSET = 0x4E00
UNICODE_OFFSET_LARGE = 0xE000


def codes_to_chars(codes, codebook_size: int, copy_before_conversion: bool = True, unicode_offset: int = UNICODE_OFFSET_LARGE) -> str:
    """codes [num_codebooks, T] (numpy / torch / list) -> str of length num_codebooks*T (frame-major)."""
    if hasattr(codes, "detach"):
        codes = codes.detach().cpu().numpy()
    codes = np.asarray(codes)
    if codes.ndim == 1:
        codes = codes[None, :]
    if codes.ndim != 2:
        raise ValueError("codes must be [num_codebooks, T]")
    if codes.size and (codes.min() < 0 or codes.max() >= codebook_size):
        raise ValueError("code out of range")
    offs = (np.arange(codes.shape[0], dtype=np.int64) * codebook_size + unicode_offset)[:, None]
    cps = (codes.astype(np.int64) + offs).T.reshape(-1)
    # one vectorised conversion instead of a chr() per code (this runs on every frame of the duplex loop); "surrogatepass" keeps
    # the code points of the small offset that fall into U+D800..DFFF, which chr() also produces
    return cps.astype("<u4").tobytes().decode("utf-32-le", "surrogatepass")


def chars_to_codes(chars: str, num_codebooks: int, codebook_size: int, return_tensors: Union[str, None] = None,
                   unicode_offset: int = UNICODE_OFFSET_LARGE):
    """str -> codes [num_codebooks, T]; trailing chars that do not fill a frame are dropped."""
    cps = np.frombuffer(chars.encode("utf-32-le", "surrogatepass"), dtype="<u4").astype(np.int64)
    T = len(cps) // num_codebooks
    cps = cps[: T * num_codebooks].reshape(T, num_codebooks).T
    offs = (np.arange(num_codebooks, dtype=np.int64) * codebook_size + unicode_offset)[:, None]
    codes = cps - offs
    if codes.size and (codes.min() < 0 or codes.max() >= codebook_size):
        raise ValueError("character outside the codec range")
    if return_tensors == "pt":
        import torch
        return torch.from_numpy(np.ascontiguousarray(codes))
    return codes
