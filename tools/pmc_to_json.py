#!/usr/bin/env python3
"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_traffic.json.
usage: pmc_to_json.py <fetch_counter_collection.csv> <write_counter_collection.csv> <B> <out.json>
FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a wide coalesced stream -> doubled
(/opt/skills/guides/MI355X_MICROARCH.md, section HBM)."""
import collections
import csv
import json
import os
import sys

fetch_csv, write_csv, b, out = sys.argv[1:5]


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "vsel::" in n:
            agg[n.split("(")[0].replace("void ", "").split("<")[0].replace("vsel::", "")].append(float(r["Counter_Value"]))
    return {k: max(v) for k, v in agg.items()}      # max = whole-batch launches (halves of the pipeline read half)


f, w = per_kernel(fetch_csv), per_kernel(write_csv)
tab = json.load(open(out)) if os.path.exists(out) else {}
tab[str(b)] = {k: {"fetch_size_kb_raw": f[k], "write_size_kb_raw": w.get(k, 0.0),
                   "hbm_bytes": int((2.0 * f[k] + w.get(k, 0.0)) * 1024)} for k in f}
json.dump(tab, open(out, "w"), indent=1, sort_keys=True)
print(json.dumps(tab[str(b)], indent=1))
