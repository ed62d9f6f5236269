#!/usr/bin/env python3
"""Trim a rocprofv3 --stats kernel_stats.csv to the libvsel kernels (+ an 'other' line) for profiles/."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
keep = [r for r in rows if "vsel::" in r["Name"]]
other = [r for r in rows if "vsel::" not in r["Name"]]
tot = sum(int(r["TotalDurationNs"]) for r in keep) or 1
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev", "PctOfVselTime"])
    for r in keep:
        name = r["Name"].split("(")[0].replace("void ", "")
        w.writerow([name, r["Calls"], r["TotalDurationNs"], f'{float(r["AverageNs"]):.1f}', r["MinNs"], r["MaxNs"],
                    f'{float(r["StdDev"]):.1f}', f'{100.0 * int(r["TotalDurationNs"]) / tot:.2f}'])
    w.writerow(["(non-vsel kernels: torch RNG / copies during setup)", sum(int(r["Calls"]) for r in other),
                sum(int(r["TotalDurationNs"]) for r in other), "", "", "", "", ""])
print(open(dst).read())
