#!/usr/bin/env python3
"""Where does the 64-rows-per-wave dQ pass beat attn_bwd_dq_kernel?  Per-kernel times (HIP events of the library's profiler), alternating."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N, ops
shapes = [(8, 524), (32, 524), (8, 1100), (16, 1100), (1, 2368), (4, 2368), (16, 2368), (1, 4096), (4, 4096), (16, 4096), (2, 8192)]
for nseq, L in shapes:
    g = torch.Generator(device="cuda").manual_seed(7)
    T = nseq * L
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
    res = {0: [], 1: []}
    for rnd in range(2):
        for dq64 in (0, 1):
            with N.debug_knob(attn_bwd_dq64=dq64):
                for _ in range(10):
                    ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                N.profile_start()
                for _ in range(10):
                    ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                prof = N.profile_stop()
            name = "attn_bwd_dq64_kernel" if dq64 else "attn_bwd_dq_kernel"
            res[dq64].append(round(prof[name][0] / prof[name][1] * 1e3, 1))
    a, b = min(res[0]), min(res[1])
    print(json.dumps({"n_seq": nseq, "L": L, "dq_us": a, "dq64_us": b, "ratio": round(a / b, 3)}), flush=True)
