#!/usr/bin/env python3
"""Per-kernel duration AND the idle gap in front of each kernel from a rocprofv3 --kernel-trace CSV:
    python tools/trace_gaps.py <..._kernel_trace.csv> [skip_first_n_calls]
Only libvsel kernels are listed; a 'call' = the kernel sequence between two colsum_partial launches."""
import csv
import sys
from collections import OrderedDict, defaultdict

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        if "vsel::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vsel::", "")
            name = name.split("<")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 5
first = rows[0][2]
calls, cur = [], []
for r in rows:
    if r[2] == first and cur:
        calls.append(cur)
        cur = []
    cur.append(r)
calls.append(cur)
calls = calls[skip:]
dur, gap, wgs = defaultdict(list), defaultdict(list), OrderedDict()
span = []
for c in calls:
    span.append((c[-1][1] - c[0][0]) / 1e3)
    for i, (s, e, nm, g) in enumerate(c):
        key = f"{i:02d} {nm}"
        wgs[key] = g
        dur[key].append((e - s) / 1e3)
        gap[key].append((s - c[i - 1][1]) / 1e3 if i else 0.0)
call_gap = [(calls[i + 1][0][0] - calls[i][-1][1]) / 1e3 for i in range(len(calls) - 1)]
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")  # noqa: E731
print(f"{len(calls)} calls; median span first-start..last-end {med(span):.2f} us; median gap between calls {med(call_gap):.2f} us")
print(f"{'kernel':46s} {'wgs':>7s} {'dur_us':>9s} {'gap_before_us':>14s}")
td = tg = 0.0
for key in wgs:
    d_, g_ = med(dur[key]), med(gap[key])
    td += d_
    tg += g_
    print(f"{key:46s} {wgs[key]:7d} {d_:9.2f} {g_:14.2f}")
print(f"{'sum':46s} {'':7s} {td:9.2f} {tg:14.2f}")
