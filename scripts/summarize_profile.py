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
