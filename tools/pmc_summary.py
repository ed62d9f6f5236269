#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv files per (kernel, counter): usage pmc_summary.py <dir-or-csv>..."""
import collections
import csv
import glob
import os
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for arg in sys.argv[1:]:
    paths = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for p in paths:
        for r in csv.DictReader(open(p)):
            n = r["Kernel_Name"]
            if "vsel::" not in n:
                continue
            n = n.split("(")[0].replace("void ", "").replace("vsel::", "")
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:32s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
