cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for a in 0 1 2 3 4 7; do
  RCA_CONV_ABLATE=$a timeout -k 10 120 python3 $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-duplex > $R/gpurun_out/abl_$a.log 2>&1
  python3 - <<PY
import json
l=[x for x in open("$R/gpurun_out/abl_$a.log") if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']
print("abl=$a ms/step=%.3f conv TF=%.1f avg_launch_ms=%.3f"%(d['ms_per_step'], r['achieved'], r['avg_launch_ms']))
PY
done
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc0 -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-duplex > $R/gpurun_out/pmc0.log 2>&1
ls $R/gpurun_out/pmc0/*/ 
