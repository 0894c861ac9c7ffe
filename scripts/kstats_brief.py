"""Per-kernel averages of the LM step kernels from a rocprofv3 --stats output directory (used by lm_gemv_sweep.sh)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")
if not f:
    print("   (no kernel_stats.csv)")
    sys.exit(0)
tot = 0.0
for r in csv.DictReader(open(f[0])):
    n = r["Name"]
    m = re.search(r"(lm_gemv_kernel<[^>]*>|lm_attn_mfma_combine_kernel|lm_attn_mfma_kernel|lm_frame_\w+|samp_\w+_kernel|lm_embed_kernel)", n)
    if m:
        print(f"   {m.group(1):42s} calls {int(r['Calls']):6d} avg_us {float(r['AverageNs']) / 1e3:8.2f}  total_ms {float(r['TotalDurationNs']) / 1e6:8.2f}")
