cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in 0 1 2; do
  RCA_CONV_PRIO=$a timeout -k 10 120 python3 $R/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-duplex > $R/gpurun_out/prio_$a.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$R/gpurun_out/prio_$a.log") if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']
print("prio=$a ms/step=%.3f conv TF=%.1f avg_launch_ms=%.3f"%(d['ms_per_step'], r['achieved'], r['avg_launch_ms']))
PY
done
