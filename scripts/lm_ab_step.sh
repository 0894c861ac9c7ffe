# step time of the in-tree library against another build (RCA_AB_LIB), bf16 / q8_0 / q4_k at 6.6 k and 2.2 k context, unprofiled,
# then the sampler kernels' averages under rocprofv3.  usage (GPU box): RCA_AB_LIB=scripts/dbg/librca_hip_head.so bash scripts/lm_ab_step.sh
cd $GRAFT_REPO_ROOT
for fmt in bf16 q8_0 q4_k; do
  for ctx in 6600 2165; do
    for lib in "${RCA_AB_LIB:-}" ""; do
      ( [ $fmt != bf16 ] && export RCA_LM_FORMAT=$fmt; [ -n "$lib" ] && export RCA_LIB_PATH=$GRAFT_REPO_ROOT/$lib; echo -n "${lib:-in-tree} : "; python3 scripts/lm_profile.py $ctx 300 | tail -1 )
    done
  done
done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lmab
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lmab -- python3 $GRAFT_REPO_ROOT/scripts/lm_profile.py 6600 100 > /tmp/lmab.txt 2>&1
python3 $GRAFT_REPO_ROOT/scripts/kstats.py /tmp/lmab 60 | grep -E "samp_|1, 1, 16, 1, 0" | cut -c1-70,93-120
