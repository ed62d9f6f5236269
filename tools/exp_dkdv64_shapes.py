#!/usr/bin/env python3
"""Where does the one-wave-per-SIMD dK / dV pass (in either item form) beat the 8-wave attn_bwd_dkdv_kernel in the form the library picks
for it (lib: per-q-head split form on few items, else q heads inside the item)?  Per-kernel times (HIP events of the library's profiler), alternating; the split form's reduce kernel is counted with it."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N, ops
shapes = [(8, 524), (32, 524), (64, 524), (8, 1100), (16, 1100), (32, 1100), (1, 2368), (4, 2368), (16, 2368), (1, 4096), (4, 4096),
          (16, 4096), (1, 8192), (2, 8192), (4, 8192)]
DKDV = ("attn_bwd_dkdv_kernel", "attn_bwd_dkdv64_kernel", "attn_bwd_group_sum_kernel")
for nseq, L in shapes:
    g = torch.Generator(device="cuda").manual_seed(7)
    T = nseq * L
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
    res = {"lib": [], "nosplit": [], "dkdv64": [], "dkdv64_split": []}
    kn = {"lib": dict(attn_bwd_dkdv64=0), "nosplit": dict(attn_bwd_dkdv64=0, attn_bwd_split=0), "dkdv64": dict(attn_bwd_dkdv64=1, attn_bwd_split=0),
          "dkdv64_split": dict(attn_bwd_dkdv64=1, attn_bwd_split=1)}
    for rnd in range(2):
        for name, kw in kn.items():
            with N.debug_knob(**kw):
                for _ in range(6):
                    ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                N.profile_start()
                for _ in range(6):
                    ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                prof = N.profile_stop()
            us = sum(prof[n][0] / prof[n][1] * 1e3 for n in prof if any(n.startswith(d) for d in DKDV))
            res[name].append(round(us, 1))
    r = {n: min(x) for n, x in res.items()}
    print(json.dumps({"n_seq": nseq, "L": L, **r, "ratio_vs_lib": round(r["lib"] / min(r["dkdv64"], r["dkdv64_split"]), 3)}), flush=True)
