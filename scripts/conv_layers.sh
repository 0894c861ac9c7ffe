# Per-kernel averages of the batch-encode step (rocprofv3 kernel trace).  Run on the GPU box: bash scripts/conv_layers.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/convprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/convprof -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-duplex --no-trim-leg --no-bf16-leg --no-cli-leg > $R/gpurun_out/conv_layers.log 2>&1
grep "^{" $R/gpurun_out/conv_layers.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],3), 'TF', round(d['roofline']['achieved'],1))"
python3 $R/scripts/kstats.py /tmp/convprof 12
