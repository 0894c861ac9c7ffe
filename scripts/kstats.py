"""Print the top rows of a rocprofv3 kernel_stats.csv found under a directory.  usage: kstats.py <dir> [rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for i, r in enumerate(csv.DictReader(open(f))):
    if i >= n:
        break
    print(f"{r['Name'].replace('void ', '')[:92]:92s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} {float(r['Percentage']):6.2f}")
