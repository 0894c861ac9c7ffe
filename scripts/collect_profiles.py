"""Copies the judged summaries of a scripts/profile_round.sh run from gpurun_out/prof_<tag>/ into profiles/<tag>/.
usage: collect_profiles.py <tag>
The destination's generated files are replaced as a set (nothing of an older run survives next to a newer SUMMARY.txt), every
source file is the NEWEST match (gpurun_out/ accumulates the output of every run), the run's source hash must equal the working
tree's, and the commit is stamped into SUMMARY.txt."""
import collections, csv, glob, json, os, shutil, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from profile_common import newest, source_hash
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = f"gpurun_out/prof_{tag}", f"profiles/{tag}"
os.makedirs(dst, exist_ok=True)
summary = open(f"{src}/SUMMARY.txt").read()
run_hash = summary.split("source_hash=")[1].split()[0] if "source_hash=" in summary else None
if run_hash != source_hash():
    sys.exit(f"profile run was made from sources {run_hash}, the working tree is {source_hash()}: re-run scripts/profile_round.sh {tag}")
for f in glob.glob(f"{dst}/*_kernel_stats.csv") + glob.glob(f"{dst}/*_stdout.log") + glob.glob(f"{dst}/pmc_*_per_kernel_avg.csv") + glob.glob(f"{dst}/traffic.json"):
    os.remove(f)
head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True).stdout.strip()
dirty = subprocess.run(["git", "status", "--porcelain", "--", "bench.py", "realtime_codec_agent_amd", "include"], capture_output=True, text=True).stdout.strip()
open(f"{dst}/SUMMARY.txt", "w").write(f"git_head={head}{' +uncommitted changes' if dirty else ''}\n" + summary)
for sub in ("bench", "bench_bf16", "lm", "lm_6k", "lm_q8", "lm_q4k"):
    ks = newest(f"{src}/{sub}/*/*kernel_stats.csv")
    if not ks:
        continue
    shutil.copy(ks, f"{dst}/{sub}_kernel_stats.csv")
    lines = [l for l in open(f"{src}/{sub}_stdout.log") if l.startswith("{") or l.startswith("ctx=") or l.startswith("fmt=")]
    open(f"{dst}/{sub}_stdout.log", "w").writelines(lines)
if os.path.exists(f"{src}/lm_6k_step_timeline.txt"):
    shutil.copy(f"{src}/lm_6k_step_timeline.txt", f"{dst}/lm_6k_step_timeline.txt")
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    cc = newest(f"{src}/{sub}/*/*counter_collection.csv")
    if not cc:
        continue
    rows = list(csv.DictReader(open(cc)))
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
def _avg(sub, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(newest(f"{src}/{sub}/*/*counter_collection.csv")))
            if r["Counter_Name"] == counter and "conv1d_mfma_kernel" in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)
if newest(f"{src}/pmc_fetch/*/*counter_collection.csv") and newest(f"{src}/pmc_write/*/*counter_collection.csv"):
    f, nf = _avg("pmc_fetch", "FETCH_SIZE")
    w, nw = _avg("pmc_write", "WRITE_SIZE")
    json.dump({"conv1d_mfma_kernel": {"hbm_bytes_per_launch": (2 * f + w) * 1024, "fetch_kib_avg": f, "write_kib_avg": w, "dispatches": [nf, nw],
                                      "source_hash": run_hash, "git_head": head,
                                      "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of bench.py --steps 4; FETCH_SIZE x2 (MI355X_MICROARCH.md, HBM)"}},
              open(f"{dst}/traffic.json", "w"), indent=1)
print("collected into", dst, sorted(os.listdir(dst)))
