"""Condenses a scripts/profile_round.sh output directory into a text summary (committed under profiles/)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]


def short(n):
    n = n.replace("void ", "")
    return n[:78]


def stats(sub, top=14):
    fs = glob.glob(f"{d}/{sub}/*/*kernel_stats.csv")
    if not fs:
        print(f"[{sub}] no kernel_stats.csv")
        return
    print(f"== {sub}: rocprofv3 --kernel-trace --stats (per kernel: calls, avg us, % of GPU time)")
    for i, r in enumerate(csv.DictReader(open(fs[0]))):
        if i >= top:
            break
        print(f"{short(r['Name']):78s} {int(r['Calls']):7d} {float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}")


def pmc(sub):
    fs = glob.glob(f"{d}/{sub}/*/*counter_collection.csv")
    ts = glob.glob(f"{d}/{sub}/*/*kernel_trace.csv")
    if not fs:
        print(f"[{sub}] no counter_collection.csv")
        return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    if ts:
        for r in csv.DictReader(open(ts[0])):
            dur[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k, v in agg.items():
        if not any(s in k for s in ("conv", "vq_", "in_proj", "lm_gemv", "lm_attn")):
            continue
        out[k] = ({c: sum(x) / len(x) for c, x in v.items()}, (sum(dur[k]) / len(dur[k])) if dur[k] else 0.0)
    return out


stats("bench")
stats("lm")
stats("lm_q8")
sq = pmc("pmc_sq")
if sq:
    print("== pmc_sq: per-dispatch averages (profiled run; SQ_* are per-XCD sums, quad-cycles for WAVE/WAIT)")
    for k, (c, du) in sq.items():
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
