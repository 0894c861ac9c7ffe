set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab
for fmt in bf16 q8_0 q4_k; do
  arg=$fmt; [ $fmt = bf16 ] && arg=""
  RCA_LIB_PATH=$GRAFT_REPO_ROOT/scripts/dbg/librca_hip_head.so timeout -k 10 500 python scripts/ab_logits.py $arg > gpurun_out/ab/a_$fmt.txt 2>gpurun_out/ab/a_$fmt.err
  timeout -k 10 500 python scripts/ab_logits.py $arg > gpurun_out/ab/b_$fmt.txt 2>gpurun_out/ab/b_$fmt.err
  if diff -q gpurun_out/ab/a_$fmt.txt gpurun_out/ab/b_$fmt.txt; then echo "$fmt: identical ($(wc -l < gpurun_out/ab/b_$fmt.txt) lines)"; else echo "$fmt: DIFFERENT"; diff gpurun_out/ab/a_$fmt.txt gpurun_out/ab/b_$fmt.txt | head -6; fi
done
