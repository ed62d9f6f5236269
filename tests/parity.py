"""Parity bookkeeping shared by the -m gpu tests: error measures in bf16 ulps and a JSONL log of the observed margins.

north_star states "attention outputs within 1e-3 bf16".  A bf16 STORE alone is off by up to 2^-9 = 1.95e-3 relative, so
the comparison is made where that statement is meaningful: the kernel's bf16 output against the fp64 oracle ROUNDED TO
bf16, counted in bf16 ulps (1 ulp = 2^-8 relative).  The gated ulp is taken at the scale of the element's output ROW
(largest |ref| of that (token, head) vector): an attention output is a convex combination of V rows whose probabilities
are rounded to bf16 before P.V (as flash-attn does), so its error is bounded by 2^-9 * max|V| -- it scales with the
summands, not with an element that happens to cancel towards 0.  Also logged, not gated: ulps at the element's own
magnitude and with the floor rowmax / 8, the fraction of bit-equal elements, max / mean |err| against the un-rounded oracle.
Every call appends the observed numbers to gpurun_out/parity/<round>.jsonl (merged into profiles/ by tools/merge_parity.py).
"""
from __future__ import annotations

import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "gpurun_out", "parity", "r02_parity.jsonl")

MAX_ULP_FWD = 2.0        # gate: every element within 2 bf16 ulps (ulp at its row's scale) of the bf16-rounded oracle
MEAN_ABS_FWD = 1e-3      # gate: mean |err| vs the un-rounded fp64 oracle


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp64 -> nearest bf16 (ties to even) -> fp64."""
    f = np.asarray(x, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64)


def bf16_ulp(mag: np.ndarray) -> np.ndarray:
    """Spacing of bf16 numbers at magnitude `mag` (8 significand bits): 2^(floor(log2 mag) - 7)."""
    mag = np.maximum(np.asarray(mag, dtype=np.float64), 2.0 ** -100)
    return 2.0 ** (np.floor(np.log2(mag)) - 7)


def fwd_metrics(got: np.ndarray, ref64: np.ndarray) -> dict:
    """got: kernel output (bf16 values as float), ref64: fp64 oracle; trailing axis = head_dim."""
    got = np.asarray(got, dtype=np.float64)
    ref64 = np.asarray(ref64, dtype=np.float64)
    rb = bf16_round(ref64)
    d = np.abs(got - rb)
    rowmax = np.abs(rb).max(axis=-1, keepdims=True)
    out = {"max_abs_err": float(np.abs(got - ref64).max()), "mean_abs_err": float(np.abs(got - ref64).mean()),
           "max_abs_ref": float(np.abs(ref64).max()), "n": int(got.size),
           "frac_bit_equal": float((d == 0).mean())}
    for tag, floor in (("elem", 0.0), ("row8", 1.0 / 8), ("row", 1.0)):
        ulps = d / bf16_ulp(np.maximum(np.abs(rb), floor * rowmax))
        out[f"max_ulp_{tag}"] = float(ulps.max())
        if tag == "row8":
            out["frac_within_1ulp"] = float((ulps <= 1.0).mean())
    return out


def grad_metrics(got: np.ndarray, ref64: np.ndarray) -> dict:
    got = np.asarray(got, dtype=np.float64)
    ref64 = np.asarray(ref64, dtype=np.float64)
    scale = max(float(np.abs(ref64).max()), 1e-3)       # floor: a single-key row has dS == 0 exactly
    e = np.abs(got - ref64)
    return {"max_abs_err_over_max": float(e.max() / scale), "mean_abs_err_over_max": float(e.mean() / scale),
            "max_abs_ref": scale, "n": int(got.size)}


def record(case: str, kind: str, metrics: dict) -> None:
    try:
        os.makedirs(os.path.dirname(LOG), exist_ok=True)
        with open(LOG, "a") as f:
            f.write(json.dumps({"case": case, "kind": kind, **metrics}) + "\n")
    except OSError:
        pass


def check_fwd(case: str, got: np.ndarray, ref64: np.ndarray, max_ulp: float = MAX_ULP_FWD) -> dict:
    m = fwd_metrics(got, ref64)
    record(case, "attn_fwd", m)
    assert m["max_ulp_row"] <= max_ulp, (case, m)
    assert m["mean_abs_err"] <= MEAN_ABS_FWD * max(1.0, m["max_abs_ref"]), (case, m)
    return m
