"""Per (kernel, grid) average durations from a rocprofv3 kernel_trace.csv.  usage: ktrace_by_grid.py <dir> [rows]"""
import collections, csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"].replace("void ", "")[:46], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:n]:
    print(f"{k[0]:46s} grid=({k[1]},{k[2]},{k[3]}) n={len(v):5d} avg={sum(v)/len(v):8.1f} share={100*sum(v)/tot:5.1f}%")
