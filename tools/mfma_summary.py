#!/usr/bin/env python3
"""MFMA counters of the libvsel kernels from rocprofv3 --pmc runs (one directory per batch size) -> JSON on stdout.
MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); flops = SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512."""
import collections
import csv
import glob
import json
import sys

out = {}
for d in sys.argv[1:]:
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    t = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    if not f:
        continue
    dur = collections.defaultdict(list)
    if t:
        for r in csv.DictReader(open(t[0])):
            if "vsel::" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("vsel::", "")
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if "vsel::" in r["Kernel_Name"]:
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("vsel::", "")
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tab = {}
    for k, c in agg.items():
        m = {n: sorted(v)[len(v) // 2] for n, v in c.items()}
        e = {"median_us": round(sorted(dur[k])[len(dur[k]) // 2], 2) if dur.get(k) else None, **{n: m[n] for n in sorted(m)}}
        if e["median_us"]:
            # SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles (32 per v_mfma_f32_32x32x16_bf16); the chip offers 1024 SIMDs x the
            # kernel's duration x 2.4 GHz (max clock: a lower bound of the true utilisation)
            e["mfma_busy_frac_of_chip"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (e["median_us"] * 2400.0 * 1024)
        e["mfma_bf16_flops"] = m.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512
        tab[k] = e
    out[d.rstrip("/").split("/")[-1]] = tab
print(json.dumps(out, indent=1))
