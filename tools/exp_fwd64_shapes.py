#!/usr/bin/env python3
"""Where does the 64-rows-per-wave forward beat the 4- / 8-wave forms?  Same process, knob attn_rows64 = 0 / 1, alternating."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native, ops
shapes = [(1, 2368), (4, 2368), (8, 1100), (16, 1100), (1, 4096), (2, 4096), (4, 4096), (1, 8192), (2, 8192), (32, 524), (64, 524), (8, 2048), (16, 2368), (16, 4096)]
for nseq, L in shapes:
    g = torch.Generator(device="cuda").manual_seed(7)
    T = nseq * L
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    fl = 4.0 * L * L * 28 * 128 / 2 * nseq
    res = {0: [], 1: []}
    for rnd in range(3):
        for r64 in (0, 1):
            with _native.debug_knob("attn_rows64", r64):
                n = max(10, int(3e-3 / (fl / 1e15)))
                for _ in range(n):
                    ops.varlen_attn(q, k, v, cu, L)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    ops.varlen_attn(q, k, v, cu, L)
                e1.record()
                torch.cuda.synchronize()
                res[r64].append(round(e0.elapsed_time(e1) / n * 1e3, 1))
    a, b = min(res[0]), min(res[1])
    print(json.dumps({"n_seq": nseq, "L": L, "us_other_forms": a, "us_rows64": b, "ratio": round(a / b, 3), "TFLOPs_rows64": round(fl / b / 1e6, 1)}), flush=True)
