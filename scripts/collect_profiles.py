"""Copies the judged summaries of a scripts/profile_round.sh run from gpurun_out/prof_<tag>/ into profiles/<tag>/.
usage: collect_profiles.py <tag>"""
import collections, csv, glob, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = f"gpurun_out/prof_{tag}", f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
shutil.copy(f"{src}/SUMMARY.txt", f"{dst}/SUMMARY.txt")
for sub in ("bench", "lm", "lm_q8"):
    if not glob.glob(f"{src}/{sub}/*/*kernel_stats.csv"):
        continue
    shutil.copy(glob.glob(f"{src}/{sub}/*/*kernel_stats.csv")[0], f"{dst}/{sub}_kernel_stats.csv")
    lines = [l for l in open(f"{src}/{sub}_stdout.log") if l.startswith("{") or l.startswith("ctx=") or l.startswith("fmt=")]
    open(f"{dst}/{sub}_stdout.log", "w").writelines(lines)
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    rows = list(csv.DictReader(open(glob.glob(f"{src}/{sub}/*/*counter_collection.csv")[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(f"{dst}/{sub}_per_kernel_avg.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "avg_value"])
        for k, cs in sorted(agg.items()):
            for c, v in sorted(cs.items()):
                w.writerow([k[:120], c, len(v), sum(v) / len(v)])
# HBM bytes per launch of the dominant kernel (bench.py's roofline.traffic): FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE, KiB
# units, separate --pmc passes; averaged over every conv1d_mfma_kernel dispatch of the profiled steps
import json
def _avg(sub, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(glob.glob(f"{src}/{sub}/*/*counter_collection.csv")[0]))
            if r["Counter_Name"] == counter and "conv1d_mfma_kernel" in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)
f, nf = _avg("pmc_fetch", "FETCH_SIZE")
w, nw = _avg("pmc_write", "WRITE_SIZE")
json.dump({"conv1d_mfma_kernel": {"hbm_bytes_per_launch": (2 * f + w) * 1024, "fetch_kib_avg": f, "write_kib_avg": w, "dispatches": [nf, nw],
                                  "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of bench.py --steps 4; FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)"}},
          open(f"{dst}/traffic.json", "w"), indent=1)
print("collected into", dst, os.listdir(dst))
