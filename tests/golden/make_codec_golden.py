"""Generates tests/golden/codec_*.npz from the CPU oracle (oracle/codec_oracle.c).

The reference holds no golden vectors for the codec (its arithmetic lives in the absent
MagiCodec package), so these fixtures pin THIS build's codec definition:
seeded weights + seeded PCM -> code ids -> decoded PCM.  Run from the repo root:
    python tests/golden/make_codec_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import bench_signal, rich_signal  # noqa: E402
from oracle.codec import OracleCodec  # noqa: E402
from realtime_codec_agent_amd.codec_model import CodecConfig, init_codec_weights, tiny_codec_config  # noqa: E402


def main():
    out = os.path.dirname(os.path.abspath(__file__))
    for tag, cfg in (("tiny", tiny_codec_config()), ("full", CodecConfig())):
        oc = OracleCodec(cfg, init_codec_weights(cfg, seed=0))
        pcm = np.stack([bench_signal(32000, 0), rich_signal(32000, 5)])
        codes = oc.encode(pcm)
        rec = oc.decode(codes)
        ragged = rich_signal(4000, 7)[None, :]  # T not a hop multiple -> pad_audio
        codes_r = oc.encode(ragged)
        np.savez_compressed(
            os.path.join(out, f"codec_{tag}.npz"),
            weights_seed=0, codes=codes, codes_ragged=codes_r,
            pcm_head=rec[:, :3200], pcm_tail=rec[:, -3200:],
            pcm_sum=np.float64(rec.astype(np.float64).sum()),
            codebook_head=oc.codebook()[:64],
        )
        print(tag, codes.shape, len(np.unique(codes)), codes_r.shape)


if __name__ == "__main__":
    main()
