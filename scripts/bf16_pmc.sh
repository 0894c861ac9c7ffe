# PMC passes over the opt-in bf16 legs of bench.py (blocked bf16 pipeline): per-kernel averages of the SQ counters and the HBM
# traffic (FETCH_SIZE x2 per the gfx950 note, WRITE_SIZE; separate passes).  On the GPU box: bash scripts/bf16_pmc.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for pass in "sq GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $pass; name=$1; shift
  rm -rf /tmp/bfpmc_$name
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d /tmp/bfpmc_$name -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-duplex --no-trim-leg --no-cli-leg > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
def load(name):
    rows = list(csv.DictReader(open(glob.glob(f"/tmp/bfpmc_{name}/*/*counter_collection.csv")[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"]
        if "blk_kernel" not in k: continue
        agg[k[:56]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r: agg[k[:56]]["dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
sq, fe, wr = load("sq"), load("fetch"), load("write")
for k in sorted(sq):
    s = sq[k]
    busy = s.get("SQ_BUSY_CYCLES", 0)
    print(f"{k:56s} dur_us={s.get('dur_us', 0):7.1f} mfma_busy={4 * s.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / max(busy, 1):5.2f} wait_any={s.get('SQ_WAIT_ANY', 0) / max(s.get('SQ_WAVE_CYCLES', 1), 1):5.2f} "
          f"wait_inst={s.get('SQ_WAIT_INST_ANY', 0) / max(s.get('SQ_WAVE_CYCLES', 1), 1):5.2f} lds_conf={s.get('SQ_LDS_BANK_CONFLICT', 0):.2e} "
          f"fetch_MB(x2)={2 * fe.get(k, {}).get('FETCH_SIZE', 0) / 1024:8.1f} write_MB={wr.get(k, {}).get('WRITE_SIZE', 0) / 1024:8.1f}")
PY
