#!/bin/bash
# per-kernel times of the LM decode step (rocprofv3 --kernel-trace --stats), run on the GPU box via gpurun
# usage: bash scripts/lm_kernel_stats.sh [ctx] [steps] [eager]     (RCA_LM_FORMAT / RCA_LM_PREFETCH pass through the environment)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lmstats
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/lm_profile.py ${1:-1000} ${2:-50} $3 > $OUT/stdout.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/*/*kernel_stats.csv")[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= 18: break
    print(f"{r['Name'].replace('void ','')[:90]:90s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.1f} {float(r['Percentage']):6.2f}")
PY
tail -1 $OUT/stdout.log
