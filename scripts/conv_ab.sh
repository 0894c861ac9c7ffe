# A/B of two builds of the library on the batch-encode step (per-kernel averages from rocprofv3).  On the GPU box:
#   bash scripts/conv_ab.sh <tag> [path of the library to load]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
if [ -n "$2" ]; then export RCA_LIB_PATH=$2; fi
mkdir -p $R/gpurun_out/ab
rm -rf /tmp/convprof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/convprof_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-duplex --no-trim-leg --no-bf16-leg --no-cli-leg > $R/gpurun_out/ab/$TAG.log 2>&1
grep "^{" $R/gpurun_out/ab/$TAG.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$TAG value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'TF', round(d['roofline']['achieved'],1))" | tee $R/gpurun_out/ab/$TAG.summary
python3 $R/scripts/kstats.py /tmp/convprof_$TAG 8 | tee -a $R/gpurun_out/ab/$TAG.summary
