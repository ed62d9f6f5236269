#!/usr/bin/env python3
"""Packed short / medium sequences in the 4- / 8-wave forward: atomic work queue (attn_static = 0) against the static deal of the
heaviest-first item list (1).  Same process, alternating;
outputs must be bit-identical (placement only)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native, ops
if os.environ.get("VSEL_LIB"):            # a variant build (tools/build_variant.sh) for same-box A/B runs
    _native.LIB_PATH = os.path.abspath(os.environ["VSEL_LIB"])

def ragged(n, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, (n,), generator=g).tolist()

cases = [(f"{n}x{l}", [l] * n) for n, l in ((3, 524), (4, 524), (7, 524), (9, 524), (5, 1100), (3, 1500))]
cases += [(f"c5_{n}_{sd}", ragged(n, 131, 947, sd)) for n in (3, 4, 5, 6) for sd in (1, 2)]
cases += [(f"mix_{n}_{sd}", ragged(n, 64, 2040, sd)) for n in (2, 3) for sd in (1, 2)]
if len(sys.argv) > 1 and sys.argv[1] == "packed":        # throughput-bound packed batches
    cases = [(f"{n}x{l}", [l] * n) for n, l in ((8, 524), (16, 524), (32, 524), (64, 524), (64, 300), (16, 1100), (32, 1100), (16, 1900))]
    cases += [(f"c5_{n}", ragged(n, 131, 947, n)) for n in (16, 32, 64)] + [("mix_24", ragged(24, 64, 2040, 7))]
MODES = (0, 1)        # (a third column was the static deal + L2 prefetch of the next item's Q rows, removed: profiles/r04_attn_static.txt)
for name, lens in cases:
    g = torch.Generator(device="cuda").manual_seed(7)
    T, L = sum(lens), max(lens)
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    fl = sum(4.0 * l * l * 28 * 128 / 2 for l in lens)
    res = {m: [] for m in MODES}
    outs = {}
    for rnd in range(3):
        for m in MODES:
            with _native.debug_knob(attn_static=m, attn_rows64=0):
                n = max(10, int(3e-3 / (fl / 0.5e15)))
                for _ in range(n):
                    o = ops.varlen_attn(q, k, v, cu, L)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    ops.varlen_attn(q, k, v, cu, L)
                e1.record()
                torch.cuda.synchronize()
                res[m].append(round(e0.elapsed_time(e1) / n * 1e3, 1))
                outs[m] = o
    same = all(torch.equal(outs[0], outs[m]) for m in MODES)
    b = {m: min(res[m]) for m in MODES}
    print(json.dumps({"case": name, "tokens": T, "us_queue": b[0], "us_static": b[1], "ratio_static": round(b[0] / b[1], 3),
                      "TFLOPs_queue": round(fl / b[0] / 1e6, 1), "TFLOPs_best": round(fl / min(b.values()) / 1e6, 1), "bit_identical": same}), flush=True)
