"""Condenses a scripts/profile_round.sh output directory into a text summary (committed under profiles/).
Every instantiation of the kernels the bench line is about (conv1d_mfma_kernel, lm_*, samp_*, vq_*) is printed whatever its
rank; the rest is cut at the top 8.  The header carries a hash of the kernel sources + bench.py the run was made from."""
import collections
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_common import newest, source_hash

d = sys.argv[1]
OURS = ("conv1d", "conv_in", "convtr", "lm_", "samp_", "vq_", "in_proj", "embed_codes")


def short(n):
    n = n.replace("void ", "")
    return n[:78]


def stats(sub, others=8):
    f = newest(f"{d}/{sub}/*/*kernel_stats.csv")
    if not f:
        print(f"[{sub}] no kernel_stats.csv")
        return
    print(f"== {sub}: rocprofv3 --kernel-trace --stats (per kernel: calls, avg us, % of GPU time)   [{os.path.relpath(f, d)}]")
    n_other = 0
    for r in csv.DictReader(open(f)):
        ours = any(s in r["Name"] for s in OURS)
        if not ours:
            n_other += 1
            if n_other > others:
                continue
        print(f"{short(r['Name']):78s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")


def pmc(sub):
    f = newest(f"{d}/{sub}/*/*counter_collection.csv")
    if not f:
        print(f"[{sub}] no counter_collection.csv")
        return {}
    t = newest(f"{os.path.dirname(f)}/*kernel_trace.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    if t:
        for r in csv.DictReader(open(t)):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k, v in agg.items():
        if not any(s in k for s in ("conv", "vq_", "in_proj", "lm_gemv", "lm_attn")):
            continue
        out[k] = ({c: sum(x) / len(x) for c, x in v.items()}, (sum(dur[k]) / len(dur[k])) if dur[k] else 0.0)
    return out


print(f"source_hash={source_hash()}  (sha256/16 over {', '.join(__import__('profile_common').HASHED)}; scripts/collect_profiles.py adds the commit)")
for sub in ("bench", "lm", "lm_6k", "lm_q8", "lm_q4k"):
    if os.path.isdir(f"{d}/{sub}"):
        stats(sub)


def conv_layer_table():
    """Per layer of the default codec's encoder at the bench's 256 windows of 2.0 s (the step bench.py times): algorithmic FLOP =
    2 Cin KS Cout per output column (the fused first layer also carries conv_in's 2 * 7 * 32 per input sample)."""
    f = newest(f"{d}/bench/*/*kernel_stats.csv")
    if not f:
        return
    B, T = 256, 32000
    layers = [("fused conv_in + k4s2", "conv1d_mfma_kernel<4, 2,", 2.0 * 32 * 4 * 64 * B * (T // 2) + 2.0 * 7 * 32 * B * T),
              ("k8s4", "conv1d_mfma_kernel<8, 4,", 2.0 * 64 * 8 * 128 * B * (T // 8)),
              ("k10s5", "conv1d_mfma_kernel<10, 5,", 2.0 * 128 * 10 * 256 * B * (T // 40)),
              ("k16s8", "conv1d_mfma_kernel<16, 8,", 2.0 * 256 * 16 * 512 * B * (T // 320)),
              ("k3 (conv_out)", "conv1d_mfma_kernel<3, 1,", 2.0 * 512 * 3 * 256 * B * (T // 320))]
    rows = list(csv.DictReader(open(f)))
    print("== conv layers of the 256-window step (bench run, the instantiation with the most calls per layer): us per launch, algorithmic GFLOP, TFLOP/s, fraction of the 157.3 TFLOP/s f32-MFMA peak")
    tot_us = tot_fl = 0.0
    for name, pat, fl in layers:
        cand = [r for r in rows if pat in r["Name"]]
        if not cand:
            continue
        r = max(cand, key=lambda r: int(r["Calls"]))
        us = float(r["AverageNs"]) / 1e3
        tot_us += us
        tot_fl += fl
        print(f"{name:22s} {short(r['Name'])[:44]:44s} {us:9.1f} us {fl / 1e9:8.1f} GFLOP {fl / us / 1e6:7.1f} TFLOP/s  {fl / us / 1e6 / 157.3:5.3f}")
    if tot_us:
        print(f"{'sum':22s} {'':44s} {tot_us:9.1f} us {tot_fl / 1e9:8.1f} GFLOP {tot_fl / tot_us / 1e6:7.1f} TFLOP/s  {tot_fl / tot_us / 1e6 / 157.3:5.3f}")


conv_layer_table()
sq = pmc("pmc_sq")
if sq:
    print("== pmc_sq: per-dispatch averages (profiled run; SQ_* are per-XCD sums, quad-cycles for WAVE/WAIT)")
    for k, (c, du) in sorted(sq.items()):
        clk = c.get("GRBM_GUI_ACTIVE", 0) / 8 / du if du else 0
        mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
        cap = 1024 * du * clk if du else 1
        print(f"{k:78s} dur_us={du / 1e3:8.1f} clk_GHz={clk:5.2f} mfma_busy={mf / cap if cap else 0:5.2f} "
              f"wait_any={c.get('SQ_WAIT_ANY', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)):5.2f} "
              f"wait_inst={c.get('SQ_WAIT_INST_ANY', 0) / max(1, c.get('SQ_WAVE_CYCLES', 1)):5.2f} lds_conf={c.get('SQ_LDS_BANK_CONFLICT', 0):.3g}")
fe, wr = pmc("pmc_fetch"), pmc("pmc_write")
if fe or wr:
    print("== HBM traffic per dispatch (MI355X_MICROARCH.md: FETCH_SIZE counts 1/2 of wide coalesced reads on gfx950 -> doubled here; KiB units)")
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, ({}, 0))[0].get("FETCH_SIZE", 0.0)
        w = wr.get(k, ({}, 0))[0].get("WRITE_SIZE", 0.0)
        print(f"{k:78s} fetch_MB(x2)={2 * f * 1024 / 1e6:10.1f} write_MB={w * 1024 / 1e6:10.1f} total_MB={(2 * f + w) * 1024 / 1e6:10.1f}")
