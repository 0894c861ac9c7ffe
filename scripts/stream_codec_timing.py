"""Per-call latency of the streaming codec calls (host buffers), as the duplex loop makes them once per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from realtime_codec_agent_amd.codec import MagiCodecHIP
from realtime_codec_agent_amd.codec_model import CodecConfig
from realtime_codec_agent_amd.audio_tokenizer import AudioTokenizer

model = MagiCodecHIP(CodecConfig(), device="cuda:0")
hip = model.hip
rng = np.random.default_rng(0)
x = np.clip(rng.normal(0, 0.1, (1, 32000)), -1, 1).astype(np.float32)
codes = rng.integers(0, 131072, (1, 100))


def bench(fn, n=300):
    for _ in range(10):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    return f"p50 {np.percentile(ts, 50):7.1f} us  p95 {np.percentile(ts, 95):7.1f} us"


for graphs in (True, False):
    hip.set_stream_graphs(graphs)
    print(f"graphs={graphs}")
    print("  encode_tail(32000 -> 4 codes)  ", bench(lambda: hip.encode_tail(x, 4)))
    print("  decode_tail(100 codes -> 1600) ", bench(lambda: hip.decode_tail(codes, 1600)))
hip.set_stream_graphs(True)
for v in (0, 1):
    hip.set_variant(v)
    print(f"variant {v}: encode_tail", bench(lambda: hip.encode_tail(x, 4)), " decode_tail", bench(lambda: hip.decode_tail(codes, 1600)))
hip.set_variant(1)
print("full encode(32000)               ", bench(lambda: hip.encode(x)))
print("full decode(100 codes)           ", bench(lambda: hip.decode(codes)))
tok = AudioTokenizer(codec_model=model, device="cuda:0")
chunk = x[0, :1280]
for _ in range(30):
    s = tok.tokenize_audio(chunk)
print("AudioTokenizer.tokenize_audio    ", bench(lambda: tok.tokenize_audio(chunk)))
print("AudioTokenizer.detokenize_audio  ", bench(lambda: tok.detokenize_audio(s, preroll_samples=0)))
