"""Rolling z-score statistics that gate the agent's control decisions (turn taking, interruptions).

Behaviour of realtime_codec_agent/realtime_agent_stats.py:7-51, kept exactly (the agent golden recordings depend on
it): a window of the last `window_secs` of per-chunk values; a new value is first scored against the CURRENT mean /
std (which may be stale), then appended; mean and std are refreshed after every value while fewer than
`update_interval_secs` worth of chunks are held, afterwards only when the count is a multiple of that interval; the
std is the population std around the pooled mean of all components (np.std(values, mean=...) in the reference, :40),
1.0 while there is a single value.

Storage is a preallocated ring of rows (one row per chunk, one column per component) instead of a deque of tuples:
nothing is allocated per frame on the duplex path.
"""
from typing import Tuple, Union

import numpy as np

from .realtime_agent_config import RealtimeAgentConfig

Value = Union[float, Tuple[float, ...]]


class RealtimeAgentStats:
    def __init__(self, config: RealtimeAgentConfig, value_size: int = 1, window_secs: float = 20.0, update_interval_secs: float = 5.0):
        self.value_size = value_size
        self.window_chunks = int(window_secs / config.chunk_size_secs)
        self.update_interval_chunks = int(update_interval_secs / config.chunk_size_secs)
        cap = max(1, self.window_chunks)
        self._ring = np.zeros((cap, value_size), dtype=np.float64)
        self._zring = np.zeros((cap, value_size), dtype=np.float64)
        self.reset()

    def reset(self):
        self._count = 0      # rows held (<= window)
        self._head = 0       # next row to write
        self.mean = 0.0
        self.std = 1.0

    # ---- views in insertion order (oldest first), for callers that look at the window
    def _ordered(self, ring: np.ndarray) -> np.ndarray:
        if self._count < ring.shape[0]:
            return ring[: self._count]
        return np.concatenate((ring[self._head:], ring[: self._head]))

    @property
    def values(self):
        return [tuple(r) for r in self._ordered(self._ring).tolist()]

    @property
    def values_zscores(self):
        return [tuple(r) for r in self._ordered(self._zring).tolist()]

    @property
    def last_zscore(self) -> Value:
        if self._count == 0:
            return (0.0,) * self.value_size if self.value_size > 1 else 0.0
        z = self._zring[(self._head - 1) % self._ring.shape[0]]
        return tuple(z.tolist()) if self.value_size > 1 else float(z[0])

    def add_value(self, value: Value):
        row = np.asarray(value, dtype=np.float64).reshape(-1)
        if row.shape[0] != self.value_size:
            raise ValueError(f"expected {self.value_size} components, got {row.shape[0]}")
        cap = self._ring.shape[0]
        self._ring[self._head] = row
        self._zring[self._head] = (row - self.mean) / self.std   # scored against the statistics BEFORE this value
        self._head = (self._head + 1) % cap
        self._count = min(self._count + 1, cap)
        n = self._count
        if n < self.update_interval_chunks or n % self.update_interval_chunks == 0:
            held = self._ordered(self._ring)   # oldest first: the same summation order as a list of the window
            self.mean = held.mean()
            self.std = float(np.sqrt(np.mean((held - self.mean) ** 2))) if n > 1 else 1.0


class RealtimeAgentStatsCollection:
    """The three statistics the loop keeps (realtime_agent_stats.py:44-51)."""

    def __init__(self, config: RealtimeAgentConfig):
        self.ch_abs_max = RealtimeAgentStats(config, value_size=2)
        self.event_prob = RealtimeAgentStats(config)
        self.tts_interrupt_score = RealtimeAgentStats(config)

    def reset(self):
        self.ch_abs_max.reset()
        self.event_prob.reset()
        self.tts_interrupt_score.reset()
