#!/usr/bin/env python3
"""gpurun_out/parity/r04_parity.jsonl (written by the -m gpu tests through tests/parity.py) -> profiles/r04_parity.json:
observed parity margins per test case, worst case per kind, and the gates they were held to."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity", "r04_parity.jsonl")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_parity.json")
rows = [json.loads(ln) for ln in open(src) if ln.strip()]
latest = {}
for r in rows:                      # keep the last record of every (case, kind, ordinal within the case)
    key = (r["case"], r["kind"])
    latest.setdefault(key, []).append(r)
out = {"gates": {"attn_fwd": "max_ulp_row <= 1 bf16 ulp and max_ulp_row8 <= 8 vs the bf16-rounded fp64 oracle (ulp at the output row's scale); "
                             "mean_abs_err <= 1e-3 * max(1, max_abs_ref)",
                 "attn_grad": "max_abs_err_over_max <= 2^-6 (bf16 outputs, bf16-rounded P / dS operands)",
                 "lis_bf16": "vs the reference's own bf16 run (tests/golden/lisbf16_*.npz): max_abs_dscore <= 1e-3 * max(1, max|s|); "
                             "index symmetric difference <= max(2 ties, 2, 1 % of k), every disagreement within 2e-3 of the k-th score",
                 "soft_bf16": "soft mask vs the reference's bf16 _find_ts: max |dps| <= 3e-3 (bf16 stall of t + bf16 rounding of p)",
                 "lis_bwd_bf16": "vs the reference's own bf16 BACKWARD (lisbf16_*.npz topk_grad_bf16 / bwd_*): max |got - ref| / max |ref| per "
                                 "quantity <= tests/parity.BF16_BWD_TOL (2 x the worst observed margin)"},
       "summary": {}, "cases": []}
for kind in sorted({k[1] for k in latest}):
    rs = [r for (c, k), v in latest.items() if k == kind for r in v]
    if kind == "attn_fwd":
        out["summary"][kind] = {"records": len(rs), "worst_max_ulp_row": max(r["max_ulp_row"] for r in rs),
                                "worst_max_ulp_row8": max(r["max_ulp_row8"] for r in rs),
                                "worst_max_abs_err": max(r["max_abs_err"] for r in rs),
                                "worst_mean_abs_err": max(r["mean_abs_err"] for r in rs),
                                "min_frac_bit_equal_to_bf16_oracle": min(r["frac_bit_equal"] for r in rs),
                                "min_frac_within_1ulp_row8": min(r["frac_within_1ulp"] for r in rs)}
    elif kind == "lis_bf16":
        out["summary"][kind] = {"records": len(rs), "worst_max_abs_dscore": max(r["max_abs_dscore"] for r in rs),
                                "worst_symdiff_frac": max(r[f"symdiff_{t}"] / r[f"k_{t}"] for r in rs for t in ("0p1", "0p2", "0p5")
                                                          if f"k_{t}" in r),
                                "total_symdiff": sum(r[f"symdiff_{t}"] for r in rs for t in ("0p1", "0p2", "0p5") if f"k_{t}" in r)}
    elif kind == "soft_bf16":
        out["summary"][kind] = {"records": len(rs), "worst_max_abs_dps": max(r["max_abs_dps"] for r in rs),
                                "worst_abs_dts": max(r["abs_dts"] for r in rs)}
    elif kind == "lis_bwd_bf16":
        keys = sorted({k for r in rs for k in r if k not in ("case", "kind", "n")})
        out["summary"][kind] = {"records": len(rs), **{f"worst_{k}": max(r[k] for r in rs if k in r) for k in keys}}
    else:
        out["summary"][kind] = {"records": len(rs), "worst_max_abs_err_over_max": max(r["max_abs_err_over_max"] for r in rs),
                                "worst_mean_abs_err_over_max": max(r["mean_abs_err_over_max"] for r in rs)}
for (case, kind), v in sorted(latest.items()):
    worst = max(v, key=lambda r: r.get("max_ulp_row", r.get("max_abs_err_over_max", r.get("max_abs_dscore", r.get("max_abs_dps", 0.0)))))
    out["cases"].append({"case": case, "kind": kind, "records": len(v), **{k: worst[k] for k in worst if k not in ("case", "kind")}})
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out["summary"], indent=1))
