# HBM-side fetch per conv launch (FETCH_SIZE, KiB; doubled for gfx950 by the reader).  On the GPU box: bash scripts/conv_fetch.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/convfetch_$1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/convfetch_$1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-duplex --no-trim-leg --no-bf16-leg --no-cli-leg > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
rows = list(csv.DictReader(open(glob.glob("/tmp/convfetch_$1/*/*counter_collection.csv")[0])))
agg = collections.defaultdict(list)
for r in rows:
    if "conv1d_mfma" in r["Kernel_Name"]: agg[r["Kernel_Name"][:44]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print("$1", k, "fetch_MB(x2) = %.1f" % (2 * sum(v) / len(v) / 1024))
PY
