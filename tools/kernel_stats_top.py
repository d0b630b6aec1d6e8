"""Print the top rows of a rocprofv3 *kernel_stats.csv (usage: python tools/kernel_stats_top.py <dir-or-file> [n])."""
import csv, glob, os, sys
f = sys.argv[1]
if os.path.isdir(f):
    f = sorted(glob.glob(os.path.join(f, "**", "*kernel_stats.csv"), recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f'{r["Name"][:100]:100s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"]) / 1e3:9.1f} us  tot {float(r["TotalDurationNs"]) / 1e6:9.1f} ms {100 * float(r["TotalDurationNs"]) / tot:5.1f}%')
