"""Per-kernel mean of one rocprofv3 --pmc counter_collection.csv.  usage: summarize_pmc.py IN.csv OUT.csv"""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "Dispatches", "Mean_Value", "Min_Value", "Max_Value"])
    for (k, c), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, c, len(v), f"{sum(v) / len(v):.3f}", f"{min(v):.3f}", f"{max(v):.3f}"])
