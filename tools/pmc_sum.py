"""Sum rocprofv3 counter_collection CSV rows for kernels whose name contains a pattern; print per-dispatch means."""
import csv, glob, os, sys
from collections import defaultdict
d, pat = sys.argv[1], sys.argv[2]
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
tot = defaultdict(float); disp = defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        if pat in r.get("Kernel_Name", ""):
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); disp[r["Counter_Name"]].add(r["Dispatch_Id"])
if not tot:
    print("no rows for", pat, "in", files)
for k in sorted(tot):
    n = max(1, len(disp[k]))
    print(f"{k:28s} dispatches {n:4d}  mean/dispatch {tot[k]/n:16.1f}")
