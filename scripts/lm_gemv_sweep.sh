#!/bin/bash
# Launch-geometry sweep of the decode GEMVs (RCA_GEMV_<KIND>="rows per batch,batches per workgroup"), per-kernel times from
# rocprofv3 --kernel-trace --stats of scripts/lm_profile.py.  Run on the GPU box: bash scripts/lm_gemv_sweep.sh [ctx] > gpurun_out/sweep.log
R=$GRAFT_REPO_ROOT
CTX=${1:-1000}
cd /tmp && export TMPDIR=/tmp
# note: `env VAR=.. rocprofv3` would be a launcher hop in front of the profiler, which is fine -- the hop that is forbidden is one
# between rocprofv3's `--` and the program; the variables are therefore exported in a subshell instead of using env.
run() {
  tag=$1; shift
  OUT=$R/gpurun_out/sweep_$tag
  rm -rf $OUT; mkdir -p $OUT
  ( for kv in "$@"; do export "$kv"; done
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/scripts/lm_profile.py $CTX 40 > $OUT/stdout.log 2>&1 )
  echo "== $tag: $* :: $(tail -1 $OUT/stdout.log)"
  python3 $R/scripts/kstats_brief.py $OUT
  rm -rf $OUT/*/*kernel_trace.csv
}
run base
run q8 RCA_LM_FORMAT=q8_0
