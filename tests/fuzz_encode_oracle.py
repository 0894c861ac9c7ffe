"""Fuzz the batch encoder and decoder (MFMA variant) against the C oracle: random model widths (channel counts that are and are not
multiples of the kernel's K chunk, outputs narrower than a tile), random batch sizes and window lengths (aligned and
ragged, rows shorter and longer than a wave's window of columns), random layer taps, random code sequences through the decoder.  Bit-exact or it stops.
usage: tests/fuzz_encode_oracle.py [seconds] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from realtime_codec_agent_amd.codec import HipCodec
from realtime_codec_agent_amd.codec_model import CodecConfig, init_codec_weights, tiny_codec_config
from oracle.codec import OracleCodec

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
shapes = [dict(), dict(channels=(6, 10, 12, 20, 36), latent_dim=24), dict(channels=(8, 16, 32, 64, 64)), dict(channels=(32, 64, 64, 96, 128), latent_dim=32),
          dict(channels=(5, 7, 9, 11, 13), latent_dim=8)]
models = []
for i, kw in enumerate(shapes):
    cfg = tiny_codec_config(name=f"fz{i}", **kw)
    w = init_codec_weights(cfg, seed=10 + i)
    models.append((cfg, HipCodec(cfg, w, device=0), OracleCodec(cfg, w)))
full = CodecConfig()
wf = init_codec_weights(full, seed=0)
models.append((full, HipCodec(full, wf, device=0), OracleCodec(full, wf)))
t0 = time.time()
n = 0
while time.time() - t0 < budget:
    k = int(rng.integers(0, len(models)))
    cfg, hip, oc = models[k]
    big = cfg is full
    B = int(rng.integers(1, 6 if big else 80))
    T = int(rng.choice([rng.integers(1, 12 if big else 110) * 320, rng.integers(100, 4000 if big else 36000)]))
    x = np.clip(rng.normal(0, 0.2, (B, T)), -1, 1).astype(np.float32)
    hip.set_variant(1)
    want = oc.encode(x)
    got = hip.encode(x)
    assert np.array_equal(got, want), ("encode", cfg.name, B, T, int((got != want).sum()))
    if rng.random() < 0.3:
        layer = int(rng.integers(0, cfg.n_stages + 3))
        _, tw = oc.encode(x, tap_layer=layer)
        tg = hip.encode_tap(x, layer)
        assert np.array_equal(tg, tw), ("tap", cfg.name, B, T, layer)
    if rng.random() < 0.3:   # decoder: transposed convolutions on the same kernel (phase GEMMs)
        Fd = int(rng.integers(1, 8 if big else 120))
        Bd = int(rng.integers(1, 4 if big else 24))
        codes = rng.integers(0, cfg.codebook_size, (Bd, Fd)).astype(np.int64)
        dg, dw = hip.decode(codes), oc.decode(codes)
        assert dg.shape == dw.shape and np.array_equal(dg, dw), ("decode", cfg.name, Bd, Fd, float(np.abs(dg - dw).max()))
    n += 1
print(f"fuzz ok: {n} random (model, batch, length) cases bit-identical to the C oracle in {time.time() - t0:.0f} s")
