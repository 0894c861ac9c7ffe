"""Shadow KV cache: the sliding-window trim without the prefill spike.

The reference evicts old context by pointing `llm.n_tokens` back at the header and re-evaluating the whole surviving suffix
inside the frame that trims (realtime_agent_v2.py:187-190,725-733: ~6 000 tokens every 20 s of dialogue with the default 80 s /
20 s window).  On MI355X that single frame costs ~130 ms against an 80 ms budget.  But the trim is known a full trim period
ahead: after the trim at time T the next one keeps [header] + [everything from T' = trim_to + trim_by on], and most of that is
already in the sequence.  So the post-trim cache is built in advance on a weight-sharing twin of the LM (own KV cache,
workspace and low-priority stream):

    plan      header KV copied device-to-device from the live cache (it is never recomputed by the reference either), the twin's
              position set behind it
    advance   once per frame, at its END: if at least `tile` new tokens of the future suffix exist, one prefill tile is enqueued on the
              twin (asynchronously, one graph replay: it runs in the idle time before the next chunk arrives; the next frame waits
              for it before launching its own work -- a tile BESIDE a frame's latency-bound launches cost that frame 5-6 ms);
              a tail of `keep_back` tokens is never fed early
    finish    at the trim: what was fed is compared with the sequence as it is NOW (text branches and edits may have rewritten
              it: the twin is rolled back to the first difference), the rest is evaluated, and the two handles trade caches

The live handle ends up in exactly the state recompute_kv_cache(0) leaves: n_tokens = header + suffix, same KV bits (the prefill
tiles give the same bits for any split of the token sequence; tests compare with the fresh recompute).  Any object without
make_kv_shadow (llama.cpp, test fakes) simply keeps the reference behaviour.
"""
from __future__ import annotations

from typing import List, Optional


class KVShadow:
    def __init__(self, llm, tile: int = 128, keep_back: int = 16):
        self.llm = llm
        self.twin = llm.make_kv_shadow()
        self.tile = int(tile)
        self.keep_back = int(keep_back)
        self.src_pos: Optional[int] = None      # index into input_ids where the planned window starts
        self.prefix_len = 0
        self.fed: List[int] = []                # ids handed to the twin so far (in order, from src_pos on)
        self.stats = dict(planned=0, tiles=0, swaps=0, fallbacks=0, rolled_back=0)

    @property
    def planned(self) -> bool:
        return self.src_pos is not None

    def cancel(self) -> None:
        self.src_pos = None
        self.fed = []

    def plan(self, prefix_len: int, src_pos: int) -> None:
        """Start building the cache of [header (prefix_len positions)] + input_ids[src_pos:]."""
        self.twin.set_mfma_prefill(getattr(self.llm, "_mfma_prefill", True))   # the twin evaluates exactly as the live handle would
        self.twin.copy_kv_from(self.llm, prefix_len)
        self.twin.n_tokens = prefix_len
        self.prefix_len, self.src_pos, self.fed = prefix_len, src_pos, []
        self.stats["planned"] += 1

    def advance(self, input_ids: List[int], max_tiles: int = 1) -> None:
        """Feed whole tiles of the suffix that are already known (leaving the newest keep_back tokens alone)."""
        if not self.planned:
            return
        for _ in range(max_tiles):
            a = self.src_pos + len(self.fed)
            avail = len(input_ids) - self.keep_back - a
            if avail < self.tile:
                return
            chunk = input_ids[a:a + self.tile]
            self.twin.eval_async(chunk)
            self.fed.extend(chunk)
            self.stats["tiles"] += 1

    def finish(self, input_ids: List[int], src_pos: int, prefix_len: int, end: int) -> bool:
        """Make the twin hold header + input_ids[src_pos:end] and trade caches with the live handle.  False (nothing touched) when
        the plan does not match this trim -- the caller then recomputes the reference way."""
        if not self.planned or src_pos != self.src_pos or prefix_len != self.prefix_len:
            self.stats["fallbacks"] += 1
            self.cancel()
            return False
        want = input_ids[src_pos:end]
        same = 0
        n = min(len(self.fed), len(want))
        while same < n and self.fed[same] == want[same]:
            same += 1
        if same < len(self.fed):                  # the sequence was edited / rolled back under us: rewind the twin
            self.stats["rolled_back"] += 1
        self.twin.n_tokens = prefix_len + same
        rest = want[same:]
        if rest:
            self.twin.eval_async(rest)            # prefill arithmetic whatever its length; swap_kv drains both streams
        self.llm.swap_kv(self.twin)
        self.llm.n_tokens = prefix_len + len(want)
        self.stats["swaps"] += 1
        self.cancel()
        return True
