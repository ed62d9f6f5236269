#!/usr/bin/env python3
"""Atomic work queue (attn_static = 0) against the static deal (argv[1]: 1 = forced wherever there is a queue, default -1 = the library's
rule, few rounds of items only) in EVERY persistent attention kernel: forward (4-wave / 64-rows forms as
the library picks them) and the backward's dQ and dK / dV passes, per kernel (HIP events of the library's profiler), same process,
alternating.  `rounds` = work items / resident workgroups of that kernel.  Outputs must be bit-identical (placement only)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N, ops

def ragged(n, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, (n,), generator=g).tolist()

cases = [(f"{n}x{l}", [l] * n) for n, l in ((2, 524), (3, 524), (4, 524), (8, 524), (16, 524), (16, 300), (32, 300), (2, 1100), (4, 1100), (8, 1100),
                                            (1, 2368), (2, 2368), (4, 2368), (1, 3000), (1, 4096), (2, 4096), (1, 8192))]
cases += [("c5_4", ragged(4, 131, 947, 1)), ("c5_6", ragged(6, 131, 947, 2)), ("c5_8", ragged(8, 131, 947, 4))]
cases += [(f"mix2k_{n}_{sd}", ragged(n, 64, 2040, sd)) for n in (2, 3) for sd in (1, 2, 3)]
cases += [(f"mix4k_{n}_{sd}", ragged(n, 512, 4096, sd)) for n in (1, 2) for sd in (1, 2, 3)]
for name, lens in cases:
    g = torch.Generator(device="cuda").manual_seed(7)
    T, L = sum(lens), max(lens)
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
    MODE = int(sys.argv[1]) if len(sys.argv) > 1 else -1
    res, outs = {0: {}, MODE: {}}, {}
    for rnd in range(2):
        for m in (0, MODE):
            with N.debug_knob(attn_static=m):
                for _ in range(5):
                    o = ops.varlen_attn(q, k, v, cu, L)
                    grads = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                N.profile_start()
                for _ in range(8):
                    ops.varlen_attn(q, k, v, cu, L)
                    ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                prof = N.profile_stop()
            outs[m] = (o,) + tuple(grads)
            for kn, (ms, calls) in prof.items():
                us = ms / calls * 1e3
                res[m][kn] = min(res[m].get(kn, 1e30), us)
    same = all(torch.equal(a, b) for a, b in zip(outs[0], outs[MODE]))
    row = {"case": name, "tokens": T, "bit_identical": same}
    for kn in sorted(res[0]):
        if kn in res[MODE]:
            short = kn.replace("varlen_attn_fwd_kernel", "fwd4").replace("attn_bwd_", "").replace("attn_", "").replace("_kernel", "")
            row[short] = [round(res[0][kn], 1), round(res[MODE][kn], 1), round(res[0][kn] / res[MODE][kn], 3)]
    print(json.dumps(row), flush=True)
