"""Fuzz the streaming tail calls against the full-window calls (both through the C ABI; the full-window path is the
one pinned to the oracle): random window lengths (aligned and ragged), batch sizes, keep counts, device and host
entry points.  usage: fuzz_tails.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from realtime_codec_agent_amd.codec import HipCodec
from realtime_codec_agent_amd.codec_model import CodecConfig, init_codec_weights, tiny_codec_config

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
codecs = []
for cfg in (CodecConfig(), tiny_codec_config()):
    codecs.append((cfg, HipCodec(cfg, init_codec_weights(cfg, seed=0), device=0)))
t0 = time.time()
n_enc = n_dec = 0
while time.time() - t0 < budget:
    cfg, hip = codecs[int(rng.integers(0, 2))]
    B = int(rng.integers(1, 5))
    # encode
    T = int(rng.choice([rng.integers(1, 40) * 320, rng.integers(200, 36000)]))
    x = np.clip(rng.normal(0, 0.15, (B, T)), -1, 1).astype(np.float32)
    full = hip.encode(x)
    F = full.shape[1]
    keep = int(rng.integers(1, min(F, 12) + 1))
    got = hip.encode_tail(x, keep)
    assert np.array_equal(got, full[:, -keep:]), ("encode_tail", cfg.name, B, T, keep)
    dev = torch.from_numpy(x).cuda()
    out = torch.full((B, keep), -1, dtype=torch.int64, device="cuda")
    hip.encode_tail_dev(dev.data_ptr(), B, T, keep, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), full[:, -keep:]), ("encode_tail_dev", cfg.name, B, T, keep)
    n_enc += 1
    # decode
    Fd = int(rng.integers(1, 120))
    codes = rng.integers(0, cfg.codebook_size, (B, Fd))
    pcm = hip.decode(codes)
    n = int(rng.integers(1, min(Fd * 320, 4000) + 1))
    got = hip.decode_tail(codes, n)
    assert np.array_equal(got, pcm[:, -n:]), ("decode_tail", cfg.name, B, Fd, n)
    n_dec += 1
print(f"fuzz ok: {n_enc} encode cases, {n_dec} decode cases in {time.time() - t0:.0f} s (seed {seed})")
