"""xRT profilers -- the reference's definition of the real-time factor
(realtime_codec_agent/realtime_agent_profiler.py:7-63): per chunk xRT = chunk_size_secs / elapsed,
averaged over `profiler_report_interval_secs` of audio; the reported figure is the median of those
window means (:75).  Differences: a monotonic clock, an optional device synchronisation before each
timestamp (the reference takes none, so its GPU stages are under-counted), and the raw per-chunk
latencies are kept for p50/p95 reporting.
"""
import time
from typing import Callable, List, Optional, Tuple

import numpy as np

from .realtime_agent_config import RealtimeAgentConfig


class RealtimeAgentProfiler:
    def __init__(self, config: RealtimeAgentConfig, sync: Optional[Callable[[], None]] = None):
        self.config = config
        self.sync = sync
        self.reset()

    def reset(self):
        self.report_chunk_count: int = 0
        self.realtime_factor_sum: float = 0.0
        self.realtime_factor_values: List[float] = []
        self.latencies_secs: List[float] = []
        self.chunk_start: Optional[float] = None

    def log_chunk_start(self):
        if not self.config.run_profilers:
            return
        if self.sync is not None:
            self.sync()
        self.chunk_start = time.perf_counter()

    def log_chunk_end(self):
        if not self.config.run_profilers:
            return
        if self.chunk_start is None:
            raise ValueError("Chunk start time not set. Call log_chunk_start() before log_chunk_end().")
        if self.sync is not None:
            self.sync()
        elapsed = time.perf_counter() - self.chunk_start
        self.chunk_start = None
        self.latencies_secs.append(elapsed)
        self.realtime_factor_sum += self.config.chunk_size_secs / (elapsed + 1e-8)
        self.report_chunk_count += 1
        if self.report_chunk_count * self.config.chunk_size_secs >= self.config.profiler_report_interval_secs:
            self.realtime_factor_values.append(self.realtime_factor_sum / self.report_chunk_count)
            self.realtime_factor_sum = 0.0
            self.report_chunk_count = 0

    def __enter__(self):
        self.log_chunk_start()
        return self

    def __exit__(self, exc_type, exc_value, traceback):
        self.log_chunk_end()

    def median_realtime_factor(self) -> Optional[float]:
        return float(np.median(self.realtime_factor_values)) if self.realtime_factor_values else None

    def latency_percentiles_ms(self, qs=(50, 95, 99)) -> dict:
        if not self.latencies_secs:
            return {}
        a = np.asarray(self.latencies_secs) * 1e3
        return {f"p{q}": float(np.percentile(a, q)) for q in qs}


class RealtimeAgentProfilerCollection:
    NAMES = ("total", "tokenize", "detokenize", "audio_tokenize", "audio_detokenize", "lm")

    def __init__(self, config: RealtimeAgentConfig, sync: Optional[Callable[[], None]] = None):
        self.config = config
        for name in self.NAMES:
            setattr(self, f"{name}_profiler", RealtimeAgentProfiler(config, sync))

    def reset(self):
        for name in self.NAMES:
            getattr(self, f"{name}_profiler").reset()

    def summary(self) -> dict:
        out = {}
        for name in self.NAMES:
            p = getattr(self, f"{name}_profiler")
            out[name] = dict(xrt_median=p.median_realtime_factor(), **p.latency_percentiles_ms())
        return out

    def build_plot(self, ylim: Tuple[Optional[float], Optional[float]] = (0.5, 3.0)):
        """Same figure as the reference (realtime_agent_profiler.py:65-114): one line per profiler, its
        median dashed, and the 1.0 real-time threshold."""
        import matplotlib.pyplot as plt
        step = self.config.profiler_report_interval_secs
        n = len(self.total_profiler.realtime_factor_values)
        x = np.arange(1, n + 1) * step
        fig, ax = plt.subplots(figsize=(14, 4))
        for i, name in enumerate(self.NAMES):
            vals = getattr(self, f"{name}_profiler").realtime_factor_values
            if not vals:
                continue
            ax.plot(x[: len(vals)], vals, label=name, color=f"C{i}")
            ax.axhline(y=np.median(vals), xmin=0.05, xmax=0.95, color=f"C{i}", linestyle="--", linewidth=1.5, label=f"{name} (median)")
        ax.axhline(y=1.0, xmin=0.05, xmax=0.95, color="orange", linestyle="--", linewidth=2.5, label="threshold")
        ax.set_title("Realtime Factor Profile")
        ax.set_xlabel("Time (seconds)")
        ax.set_ylabel("Realtime factor")
        ax.set_ylim(*ylim)
        ax.grid(True)
        fig.legend(loc="outside center right")
        return fig
