"""CodecTokenizer -- the text/codec tokenizer object the agent holds as `resources.tokenizer`.

The reference loads a HF fast tokenizer saved next to the GGUF (realtime_agent_resources.py:34): the
Llama-3 vocabulary (128256 ids) + the sequence-grammar specials + one single-character token per
codec code, added in id order so that every id above <|end_header|> is an audio token
(realtime_agent_v2.py:345,361; train_vanilla_latest.py:557,587-589).  No tokenizer files exist
offline, so this class builds that id layout directly and offers the subset of the HF API the agent
calls: encode(text, add_special_tokens=True), decode(ids, skip_special_tokens=False),
convert_tokens_to_ids, convert_ids_to_tokens, __len__.  Text falls back to UTF-8 bytes (ids 0..255)
plus single tokens for " X" speaker labels (the agent requires " A"/" B" to be ONE id,
realtime_agent_v2.py:137-138); the Llama-3 BPE merges themselves are not reproduced.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

from .codec_chars import UNICODE_OFFSET_LARGE

GRAMMAR_SPECIALS = ["<|audio_first|>", "<|text_only|>", "<|audio_only|>", "<|text_first|>", "<|agent|>", "<|agent_voice|>",
                    "<|speaker|>", "<|audio|>", "<|end_audio|>", "<|end_header|>"]


class CodecTokenizer:
    def __init__(self, base_vocab_size: int = 128256, codebook_size: int = 131072, unicode_offset: int = UNICODE_OFFSET_LARGE,
                 specials: Sequence[str] = GRAMMAR_SPECIALS, pad_to_multiple_of: int = 8):
        if base_vocab_size < 512:
            raise ValueError("base_vocab_size must be >= 512 (256 byte ids + 52 label ids + BOS)")
        self.base_vocab_size = base_vocab_size
        self.codebook_size = codebook_size
        self.unicode_offset = unicode_offset
        self.bos_token = "<|begin_of_text|>"
        self.bos_token_id = base_vocab_size - 256  # 128000 for Llama-3
        self._tok2id: Dict[str, int] = {self.bos_token: self.bos_token_id}
        # " A".." Z", " a".." z" as single ids right after the byte range
        for i, ch in enumerate("ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz"):
            self._tok2id[" " + ch] = 256 + i
        for i, s in enumerate(specials):
            self._tok2id[s] = base_vocab_size + i
        self.codec_vocab_start = base_vocab_size + len(specials)
        self._id2tok = {v: k for k, v in self._tok2id.items()}
        n = self.codec_vocab_start + codebook_size
        self.vocab_size = (n + pad_to_multiple_of - 1) // pad_to_multiple_of * pad_to_multiple_of
        self._special_strs = sorted([s for s in self._tok2id if s.startswith("<|")], key=len, reverse=True)

    def __len__(self) -> int:
        return self.codec_vocab_start + self.codebook_size

    # ---- HF-like API
    def convert_tokens_to_ids(self, tokens: Union[str, Sequence[str]]):
        if isinstance(tokens, str):
            return self._one(tokens)
        return [self._one(t) for t in tokens]

    def _one(self, tok: str) -> Optional[int]:
        if tok in self._tok2id:
            return self._tok2id[tok]
        if len(tok) == 1:
            cp = ord(tok)
            if self.unicode_offset <= cp < self.unicode_offset + self.codebook_size:
                return self.codec_vocab_start + (cp - self.unicode_offset)
            if cp < 128:
                return cp
        return None

    def convert_ids_to_tokens(self, ids: Union[int, Sequence[int]]):
        if isinstance(ids, int):
            return self._id_to_str(ids)
        return [self._id_to_str(i) for i in ids]

    def encode(self, text: str, add_special_tokens: bool = True) -> List[int]:
        ids: List[int] = [self.bos_token_id] if add_special_tokens else []
        i, n = 0, len(text)
        lo, hi = self.unicode_offset, self.unicode_offset + self.codebook_size
        while i < n:
            ch = text[i]
            cp = ord(ch)
            if lo <= cp < hi:
                ids.append(self.codec_vocab_start + cp - lo)
                i += 1
                continue
            if ch == "<" and text.startswith("<|", i):
                for s in self._special_strs:
                    if text.startswith(s, i):
                        ids.append(self._tok2id[s])
                        i += len(s)
                        break
                else:
                    ids.append(cp)
                    i += 1
                continue
            if ch == " " and i + 1 < n and (" " + text[i + 1]) in self._tok2id:
                ids.append(self._tok2id[" " + text[i + 1]])
                i += 2
                continue
            ids.extend(ch.encode("utf-8"))
            i += 1
        return ids

    def _id_to_str(self, i: int) -> str:
        if i in self._id2tok:
            return self._id2tok[i]
        if self.codec_vocab_start <= i < self.codec_vocab_start + self.codebook_size:
            return chr(self.unicode_offset + i - self.codec_vocab_start)
        if 0 <= i < 256:
            return bytes([i]).decode("latin-1")
        return ""

    def decode(self, ids: Sequence[int], skip_special_tokens: bool = False) -> str:
        out: List[str] = []
        pending = bytearray()

        def flush():
            if pending:
                out.append(pending.decode("utf-8", errors="replace"))
                pending.clear()

        for i in ids:
            i = int(i)
            if 0 <= i < 256:
                pending.append(i)
                continue
            flush()
            s = self._id_to_str(i)
            if skip_special_tokens and s.startswith("<|"):
                continue
            out.append(s)
        flush()
        return "".join(out)
