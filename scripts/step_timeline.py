"""GPU-side timeline of one decode step from a rocprofv3 kernel trace (kernel_trace.csv under <dir>): for every replay of the step the
span first kernel start -> last kernel end, and for the tail of the step (head GEMV .. sampler) each kernel's start offset, duration
and the gap to its predecessor's end, as medians over the replays.   usage: step_timeline.py <dir>"""
import csv, glob, sys
import numpy as np
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# a step ends with samp_final_kernel; it starts with the first kernel after the previous samp_final
steps, cur = [], []
for r in rows:
    cur.append(r)
    if r[2].startswith("samp_final_kernel"):
        steps.append(cur); cur = []
steps = [s for s in steps if len(s) == len(steps[-1])][-200:]
span = np.array([s[-1][1] - s[0][0] for s in steps]) / 1e3
busy = np.array([sum(e - b for b, e, _ in s) for s in steps]) / 1e3
print(f"{len(steps)} steps of {len(steps[-1])} kernels: span median {np.median(span):.1f} us (p10 {np.percentile(span, 10):.1f}, p90 {np.percentile(span, 90):.1f}); sum of kernel durations {np.median(busy):.1f} us")
tail = 6
for k in range(len(steps[-1]) - tail, len(steps[-1])):
    dur = np.median([s[k][1] - s[k][0] for s in steps]) / 1e3
    gap = np.median([s[k][0] - s[k - 1][1] for s in steps]) / 1e3
    print(f"  {steps[-1][k][2][:60]:60s} gap before {gap:6.2f} us   duration {dur:7.2f} us")
gaps = np.array([[s[k][0] - s[k - 1][1] for k in range(1, len(s))] for s in steps]) / 1e3
print(f"  gaps between consecutive kernels: median {np.median(gaps):.2f} us, mean {gaps.mean():.2f} us, sum per step {np.median(gaps.sum(axis=1)):.1f} us")
