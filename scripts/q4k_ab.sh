cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
RCA_LIB_PATH=$GRAFT_REPO_ROOT/scripts/dbg/librca_hip_head.so python scripts/ab_logits.py q4_k > gpurun_out/ab/a_q4_k.txt 2>/dev/null
python scripts/ab_logits.py q4_k > gpurun_out/ab/b_q4_k.txt 2>/dev/null
if diff -q gpurun_out/ab/a_q4_k.txt gpurun_out/ab/b_q4_k.txt; then echo "q4_k identical ($(wc -l < gpurun_out/ab/b_q4_k.txt) lines)"; else echo DIFFERENT; fi
export RCA_LM_FORMAT=q4_k
for ctx in 6600 2165; do for lib in scripts/dbg/librca_hip_head.so ""; do ( [ -n "$lib" ] && export RCA_LIB_PATH=$GRAFT_REPO_ROOT/$lib; echo -n "${lib:-in-tree} : "; python3 scripts/lm_profile.py $ctx 300 | tail -1 | cut -c1-70 ); done; done
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/q4p && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q4p -- python3 $GRAFT_REPO_ROOT/scripts/lm_profile.py 2165 60 > /tmp/q4p.txt 2>&1
python3 $GRAFT_REPO_ROOT/scripts/kstats.py /tmp/q4p 14 | grep "lm_gemv_kernel<[12], [14]" | cut -c1-50,93-120
