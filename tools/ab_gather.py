#!/usr/bin/env python3
"""Same-process A/B of the kept-row gather forms (knob lis_gather, include/vsel_debug.h) on the headline workload:
per-kernel time of gather_rows_kernel from the library's HIP-event marks, whole-step time, bit identity of the outputs.
    python tools/ab_gather.py [--images 128] [--forms 0,82,42,53,43,44]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N, ops  # noqa: E402


def arg(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


b = int(arg("--images", 128))
forms = [int(x) for x in arg("--forms", "0,82,42,53,43,44,34,24").split(",")]
iters = int(arg("--iters", 20))
n, d, hd, k = int(arg("--n", 2304)), int(arg("--d", 3584)), int(arg("--hd", 1792)), 0
k = int(arg("--k", max(1, int(n * 0.2))))
g = torch.Generator(device="cuda").manual_seed(1234)
h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
ref = None
for rnd in range(2):                     # two passes over the forms: order effects (clock, cache state) show as a spread
    for f in forms:
        with N.debug_knob("lis_gather", f):
            for _ in range(3):
                out, idx, sc = ops.lis_select(h, wq, bq, wk, bk, k)
            torch.cuda.synchronize()
            if ref is None:
                ref = (out.clone(), idx.clone())
            same = bool(torch.equal(out, ref[0]) and torch.equal(idx, ref[1]))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.lis_select(h, wq, bq, wk, bk, k)
            e1.record()
            torch.cuda.synchronize()
            N.profile_start()
            for _ in range(iters):
                ops.lis_select(h, wq, bq, wk, bk, k)
            torch.cuda.synchronize()
            prof = N.profile_stop()
            gk = [v for kn, v in prof.items() if "gather" in kn]
            print(json.dumps({"form": f, "pass": rnd, "step_us": round(e0.elapsed_time(e1) / iters * 1e3, 1),
                              "gather_us": round(sum(ms for ms, _ in gk) / max(1, sum(c for _, c in gk)) * 1e3, 1), "identical": same}), flush=True)
