import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before any HIP library: see realtime_codec_agent_amd/_native.py)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bench_signal(n: int, seed: int = 0, sr: int = 16000) -> np.ndarray:
    """three sines (220/440/1330 Hz, amp 0.1) + N(0, 0.01), clipped."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    x = sum(0.1 * np.sin(2 * np.pi * f * t) for f in (220.0, 440.0, 1330.0)) + rng.normal(0.0, 0.01, n)
    return np.clip(x, -1.0, 1.0).astype(np.float32)


def rich_signal(n: int, seed: int = 5) -> np.ndarray:
    """Amplitude-modulated noise: exercises many different codebook entries."""
    rng = np.random.default_rng(seed)
    knots = np.arange(0, n + 800, 800)
    env = np.abs(np.interp(np.arange(n), knots, rng.normal(0, 0.3, len(knots))))
    return np.clip(rng.normal(0, 1, n) * env, -1, 1).astype(np.float32)


@pytest.fixture(scope="session")
def tiny_codec():
    from realtime_codec_agent_amd.codec_model import init_codec_weights, tiny_codec_config
    cfg = tiny_codec_config()
    return cfg, init_codec_weights(cfg, seed=0)


@pytest.fixture(scope="session")
def full_codec():
    from realtime_codec_agent_amd.codec_model import CodecConfig, init_codec_weights
    cfg = CodecConfig()
    return cfg, init_codec_weights(cfg, seed=0)


@pytest.fixture(scope="session")
def tiny_oracle(tiny_codec):
    from oracle.codec import OracleCodec
    return OracleCodec(*tiny_codec)


@pytest.fixture(scope="session")
def full_oracle(full_codec):
    from oracle.codec import OracleCodec
    return OracleCodec(*full_codec)
