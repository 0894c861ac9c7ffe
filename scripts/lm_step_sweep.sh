#!/bin/bash
# Step time of the decode step (scripts/lm_profile.py, graph replay) under launch-geometry overrides, one kernel kind at a time.
# usage (GPU box): bash scripts/lm_step_sweep.sh [ctx] [fmt]
R=$GRAFT_REPO_ROOT
CTX=${1:-6600}
FMT=$2
P=RCA_GEMV
if [ "$FMT" = "q8_0" ]; then export RCA_LM_FORMAT=q8_0; P=RCA_GEMVQ; fi
run() { ( for kv in "$@"; do export "$kv"; done; echo "$* :: $(python3 $R/scripts/lm_profile.py $CTX 100 2>&1 | tail -1)" ); }
run base=1
for g in 16,4 16,8 8,2 8,4 8,8 4,8; do run ${P}_GU=$g; done
for g in 4,2 4,4; do run ${P}_DOWN=$g; done
for g in 4,2 8,1 8,2 16,1; do run ${P}_QKV=$g; done
for g in 4,2 8,1 8,2; do run ${P}_O=$g; done
for g in 16,16 16,32 8,16 8,32; do run ${P}_HEAD=$g; done
