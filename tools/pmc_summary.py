"""Mean counter value per kernel from rocprofv3 --pmc output directories.  usage: python tools/pmc_summary.py dir [dir ...] [--match substr]"""
import collections, csv, glob, os, sys
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
match = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--match=")), "")
acc = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
last = None
for (k, c), v in sorted(acc.items()):
    if k != last:
        print(k); last = k
    print(f"    {c:32s} {sum(v) / len(v):16.0f}  (mean of {len(v)})")
