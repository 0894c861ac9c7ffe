#!/bin/bash
# SQ counters of the LM kernels of one prefill + a few decode steps (rocprofv3 --pmc, counters only), run on the GPU box via gpurun
# usage: bash scripts/lm_kernel_pmc.sh [ctx] [kernel-name substring]
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lmpmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/a -- python3 $R/scripts/lm_profile.py ${1:-6600} 4 > $OUT/a_stdout.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/b -- python3 $R/scripts/lm_profile.py ${1:-6600} 4 > $OUT/b_stdout.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in "ab":
    fs = glob.glob("$OUT/%s/*/*counter_collection.csv" % tag)
    if not fs:
        print("no counter file for pass", tag); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "${2:-flash}" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k[:80])
        for c, v in sorted(d.items()):
            print(f"   {c:32s} {v / n[(k, c)]:16.0f}   (avg of {n[(k, c)]} launches)")
PY
