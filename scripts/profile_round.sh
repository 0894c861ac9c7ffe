#!/bin/bash
# Collects the rocprofv3 evidence committed under profiles/ (run on the GPU box via gpurun).
# usage: bash scripts/profile_round.sh <tag>
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT   # a fresh directory per run: nothing of an earlier run can be picked up
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the headline bench command (batch encode + duplex leg)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 $R/bench.py --no-cpu-baseline --no-trim-leg --no-bf16-leg --no-cli-leg --duplex-secs 20 > $OUT/bench_stdout.log 2>&1
# 1b. the opt-in bf16 legs (blocked bf16 pipeline): per-kernel averages behind the bench line's roofline_bf16
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_bf16 -- python3 $R/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-duplex --no-trim-leg --no-cli-leg > $OUT/bench_bf16_stdout.log 2>&1
# 2. kernel trace + stats of LM steps alone (ctx 1000, graph replay)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lm -- python3 $R/scripts/lm_profile.py 1000 50 > $OUT/lm_stdout.log 2>&1
# 2b. the same with the decode step streaming q8_0 weights (the program after -- is python3 itself; the format goes in through the environment)
( export RCA_LM_FORMAT=q8_0; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lm_q8 -- python3 $R/scripts/lm_profile.py 1000 50 > $OUT/lm_q8_stdout.log 2>&1 )
( export RCA_LM_FORMAT=q4_k; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lm_q4k -- python3 $R/scripts/lm_profile.py 1000 50 > $OUT/lm_q4k_stdout.log 2>&1 )
# 2c. the bf16 step at the context of the bench's duplex leg (6.6 k tokens)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/lm_6k -- python3 $R/scripts/lm_profile.py 6600 50 > $OUT/lm_6k_stdout.log 2>&1
# 3. PMC pass (own run, kernel-trace only): MFMA busy, waits, clock, LDS conflicts
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-duplex --no-trim-leg --no-bf16-leg --no-cli-leg > $OUT/pmc_sq_stdout.log 2>&1
# 4. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-duplex --no-trim-leg --no-bf16-leg --no-cli-leg > $OUT/pmc_fetch_stdout.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-duplex --no-trim-leg --no-bf16-leg --no-cli-leg > $OUT/pmc_write_stdout.log 2>&1
# GPU-side timeline of the replayed 6.6 k step (span, gaps, the sampler's launches) from the raw trace, before it is dropped
python3 $R/scripts/step_timeline.py $OUT/lm_6k > $OUT/lm_6k_step_timeline.txt 2>&1
python3 $R/scripts/summarize_profile.py $OUT > $OUT/SUMMARY.txt 2>&1
# the raw per-dispatch traces of the two --stats runs are large (gpurun merges at most 64 MiB back): keep the stats tables
rm -f $OUT/bench/*/*kernel_trace.csv $OUT/bench_bf16/*/*kernel_trace.csv $OUT/lm/*/*kernel_trace.csv $OUT/lm_q8/*/*kernel_trace.csv $OUT/lm_q4k/*/*kernel_trace.csv $OUT/lm_6k/*/*kernel_trace.csv
tail -60 $OUT/SUMMARY.txt
