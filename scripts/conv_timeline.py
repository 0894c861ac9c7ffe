"""Per-wave timeline of conv1d_mfma_kernel.  Needs a library built with -DRCA_CONV_TIMELINE:
    export RCA_EXTRA_HIPCC_FLAGS=-DRCA_CONV_TIMELINE RCA_LIB_PATH=/tmp/rca_convtl.so; python -c "from realtime_codec_agent_amd import _native; _native.build()"
    RCA_CONV_TIMELINE_OUT=/tmp/tl.bin python scripts/conv_timeline.py && RCA_CONV_TIMELINE_OUT=/tmp/tl.bin python scripts/conv_timeline.py analyse
(RCA_LIB_PATH keeps the diagnostic build away from the in-tree library: _native.build refuses extra flags without it).

Runs two bench-shaped encode passes (256 windows), lets the library dump (t_entry, t_loop, t_epilogue, t_exit, HW_ID)
per wave at codec destruction and prints, per layer: prologue / chunk loop / epilogue time of a wave and the gap between
one wave leaving a hardware wave slot and the next wave entering it.  wall_clock64 ticks are 10 ns.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(path):
    from realtime_codec_agent_amd.codec import HipCodec
    from realtime_codec_agent_amd.codec_model import CodecConfig, init_codec_weights
    cfg = CodecConfig()
    hip = HipCodec(cfg, init_codec_weights(cfg, seed=0), device=0)
    C, chunk, ctx, B = 2, 1600, 32000, 256
    per = B // C
    first = 20
    n_chunks = first + 2 * per
    N = n_chunks * chunk
    audio = (0.1 * torch.randn(C, N, device="cuda:0")).contiguous()
    fpc = hip.frames_per_chunk(chunk)
    codes = torch.empty((C, per * fpc * 2), dtype=torch.int64, device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for i in range(2):
        c0 = first + i * per
        hip.encode_chunk_range_dev(audio.data_ptr(), C, N, chunk, ctx, B, c0, c0 + per, codes.data_ptr() + 8 * i * per * fpc, codes.shape[1], st)
    torch.cuda.synchronize()
    hip.close() if hasattr(hip, "close") else None
    del hip


def analyse(path):
    raw = open(path, "rb").read()
    n = int(np.frombuffer(raw[:4], np.uint32)[0])
    a = np.frombuffer(raw[4:], np.int64).reshape(-1, 8)
    a = a[a[:, 0] != 0]
    print("records", len(a))
    t0, t1, t2, t3, hw, meta = a.T[:6]
    ks, s = meta & 0xFF, (meta >> 8) & 0xFF
    start = t0.min()
    half = a[t0 > (t0.min() + t0.max()) // 2] if False else a
    for key in sorted(set(zip(ks.tolist(), s.tolist()))):
        m = (ks == key[0]) & (s == key[1])
        sub = a[m]
        # the second pass only
        T0, T1, T2, T3, HW = sub[:, 0], sub[:, 1], sub[:, 2], sub[:, 3], sub[:, 4]
        nch = (sub[:, 7] >> 32)[0]
        half = (nch + 1) // 2   # stamped chunks (the even ones)
        ld, mf, wr = (sub[:, 6] & 0xFFFFFFFF) * 0.01, (sub[:, 6] >> 32) * 0.01 / half, (sub[:, 7] & 0xFFFFFFFF) * 0.01 / half
        print(f"   per chunk ({nch} chunks): shortest MFMA block of the wave {np.median(ld):5.2f} (p10 {np.percentile(ld, 10):5.2f})  fragment reads + MFMA block {np.median(mf):5.2f} (p10 {np.percentile(mf, 10):5.2f}, p90 {np.percentile(mf, 90):5.2f})"
              f"  activation + LDS write {np.median(wr):5.2f} (p90 {np.percentile(wr, 90):5.2f}) us")
        span = (T3.max() - T0.min()) * 0.01
        pro, loop, epi = (T1 - T0) * 0.01, (T2 - T1) * 0.01, (T3 - T2) * 0.01
        slot = (HW & 0xF) | (((HW >> 4) & 3) << 4) | (((HW >> 8) & 0xFF) << 8) | ((HW >> 32) << 20)
        order = np.lexsort((T0, slot))
        sl, tt0, tt3 = slot[order], T0[order], T3[order]
        same = sl[1:] == sl[:-1]
        gap = (tt0[1:] - tt3[:-1])[same] * 0.01
        gap = gap[(gap > -50) & (gap < 200)]
        print(f"k{key[0]}s{key[1]}: waves {len(sub)}, launch span {span:8.1f} us | prologue {np.median(pro):6.2f} (p90 {np.percentile(pro, 90):6.2f})"
              f"  loop {np.median(loop):6.2f} (p90 {np.percentile(loop, 90):6.2f})  epilogue-issue {np.median(epi):5.2f} (p90 {np.percentile(epi, 90):5.2f})"
              f"  slot gap {np.median(gap) if len(gap) else float('nan'):6.2f} (p90 {np.percentile(gap, 90) if len(gap) else float('nan'):6.2f}, n={len(gap)})"
              f"  distinct slots {len(set(sl.tolist()))}")


if __name__ == "__main__":
    path = os.environ.get("RCA_CONV_TIMELINE_OUT")
    assert path, "set RCA_CONV_TIMELINE_OUT=<file>"
    if len(sys.argv) > 1 and sys.argv[1] == "analyse":
        analyse(path)
    else:
        run(path)
