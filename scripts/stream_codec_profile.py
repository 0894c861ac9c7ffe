"""Kernel-level breakdown of the streaming tail calls (run under rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd.codec import MagiCodecHIP
from realtime_codec_agent_amd.codec_model import CodecConfig
model = MagiCodecHIP(CodecConfig(), device="cuda:0")
hip = model.hip
hip.set_stream_graphs(False)
rng = np.random.default_rng(0)
x = np.clip(rng.normal(0, 0.1, (1, 32000)), -1, 1).astype(np.float32)
codes = rng.integers(0, 131072, (1, 100))
for _ in range(100):
    hip.encode_tail(x, 4)
    hip.decode_tail(codes, 1600)
