"""Rolling z-score statistics that gate the agent's control decisions.

Same behaviour as realtime_codec_agent/realtime_agent_stats.py:7-51: a window of the last
`window_secs` of per-chunk values; mean/std are refreshed while the window is shorter than
`update_interval_secs` and afterwards only every `update_interval_secs`; each new value is scored
against the CURRENT (possibly stale) mean/std before the refresh.
"""
from collections import deque
from typing import Deque, Tuple, Union

import numpy as np

from .realtime_agent_config import RealtimeAgentConfig


class RealtimeAgentStats:
    def __init__(self, config: RealtimeAgentConfig, value_size: int = 1, window_secs: float = 20.0, update_interval_secs: float = 5.0):
        self.value_size = value_size
        self.window_chunks = int(window_secs / config.chunk_size_secs)
        self.update_interval_chunks = int(update_interval_secs / config.chunk_size_secs)
        self.reset()

    def reset(self):
        self.values: Deque[Tuple[float, ...]] = deque()
        self.values_zscores: Deque[Tuple[float, ...]] = deque()
        self.mean = 0.0
        self.std = 1.0

    @property
    def last_zscore(self) -> Union[float, Tuple[float, ...]]:
        if not self.values:
            return (0.0,) * self.value_size if self.value_size > 1 else 0.0
        z = self.values_zscores[-1]
        return z if self.value_size > 1 else z[0]

    def add_value(self, value: Union[float, Tuple[float, ...]]):
        if isinstance(value, (np.ndarray, np.generic)):
            value = value.tolist()
        if isinstance(value, list):
            value = tuple(value)
        elif isinstance(value, (float, int)):
            value = (value,)
        self.values.append(value)
        self.values_zscores.append(tuple((v - self.mean) / self.std for v in value))
        if len(self.values) > self.window_chunks:
            self.values.popleft()
            self.values_zscores.popleft()
        n = len(self.values)
        if n < self.update_interval_chunks or n % self.update_interval_chunks == 0:
            arr = np.asarray(self.values, dtype=np.float64)
            self.mean = arr.mean()
            # population std around the pooled mean (np.std(values, mean=...) in the reference, :40)
            self.std = float(np.sqrt(((arr - self.mean) ** 2).mean())) if n > 1 else 1.0


class RealtimeAgentStatsCollection:
    def __init__(self, config: RealtimeAgentConfig):
        self.ch_abs_max = RealtimeAgentStats(config, value_size=2)
        self.event_prob = RealtimeAgentStats(config)
        self.tts_interrupt_score = RealtimeAgentStats(config)

    def reset(self):
        for s in (self.ch_abs_max, self.event_prob, self.tts_interrupt_score):
            s.reset()
