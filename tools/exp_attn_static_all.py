#!/usr/bin/env python3
"""A/B of one placement knob in EVERY persistent attention kernel, argv[1] = knob=a,b (default attn_static=0,-1: atomic work queue against
the library's rule, static deal on few rounds of items only; attn_static=0,1 forces the deal wherever there is a queue;
attn_skip_empty=0,1: the queue handing out every item against the counter jumping over empty runs); argv[2] = ragged: ragged cases only: forward (4-wave / 64-rows forms as
the library picks them) and the backward's dQ and dK / dV passes, per kernel (HIP events of the library's profiler), same process,
alternating.  `rounds` = work items / resident workgroups of that kernel.  Outputs must be bit-identical (placement only)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N, ops
if os.environ.get("VSEL_LIB"):            # a variant build for same-box A/B runs
    N.LIB_PATH = os.path.abspath(os.environ["VSEL_LIB"])

def ragged(n, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, (n,), generator=g).tolist()

cases = [(f"{n}x{l}", [l] * n) for n, l in ((2, 524), (3, 524), (4, 524), (8, 524), (16, 524), (16, 300), (32, 300), (2, 1100), (4, 1100), (8, 1100),
                                            (1, 2368), (2, 2368), (4, 2368), (1, 3000), (1, 4096), (2, 4096), (1, 8192))]
cases += [("c5_4", ragged(4, 131, 947, 1)), ("c5_6", ragged(6, 131, 947, 2)), ("c5_8", ragged(8, 131, 947, 4))]
cases += [(f"mix2k_{n}_{sd}", ragged(n, 64, 2040, sd)) for n in (2, 3) for sd in (1, 2, 3)]
cases += [(f"mix4k_{n}_{sd}", ragged(n, 512, 4096, sd)) for n in (1, 2) for sd in (1, 2, 3)]
if len(sys.argv) > 2 and sys.argv[2] == "ragged":
    cases = [("16x524", [524] * 16), ("32x524", [524] * 32), ("16x1100", [1100] * 16), ("8x2000", [2000] * 8)]
    cases += [(f"c5_{n}", ragged(n, 131, 947, n)) for n in (12, 16, 32, 64)] + [(f"mix2k_{n}", ragged(n, 64, 2040, n)) for n in (8, 16, 24)]
    cases += [(f"mix4k_{n}", ragged(n, 256, 4096, n)) for n in (6, 12)] + [("long_short", [4000] + [200] * 40), ("two_classes", [300] * 30 + [1800] * 6)]
for name, lens in cases:
    g = torch.Generator(device="cuda").manual_seed(7)
    T, L = sum(lens), max(lens)
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
    KNOB, vals = (sys.argv[1] if len(sys.argv) > 1 else "attn_static=0,-1").split("=")
    BASE, MODE = (int(x) for x in vals.split(","))
    res, outs = {BASE: {}, MODE: {}}, {}
    for rnd in range(2):
        for m in (BASE, MODE):
            with N.debug_knob(**{KNOB: m}):
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
    same = all(torch.equal(a, b) for a, b in zip(outs[BASE], outs[MODE]))
    row = {"case": name, "tokens": T, "bit_identical": same}
    for kn in sorted(res[BASE]):
        if kn in res[MODE]:
            short = kn.replace("varlen_attn_fwd_kernel", "fwd4").replace("attn_bwd_", "").replace("attn_", "").replace("_kernel", "")
            row[short] = [round(res[BASE][kn], 1), round(res[MODE][kn], 1), round(res[BASE][kn] / res[MODE][kn], 3)]
    print(json.dumps(row), flush=True)
