#!/usr/bin/env python3
"""Training LIS block timing on one MI355X (BASELINE config 3 geometry): forward (scores + soft top-k + mask apply +
constraint mask + BCE) and backward (closed-form scorer gradients), per micro-batch, HIP-event timed."""
import json
import sys

import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native  # noqa: E402
if "--lib" in sys.argv:
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from visionselector_amd import ops  # noqa: E402

d, hd = 3584, 1792
res = {}
for n in ((int(sys.argv[sys.argv.index("--n") + 1]),) if "--n" in sys.argv else (1024, 2304, 8192)):
    k = int(n * 0.2)
    g = torch.Generator(device="cuda").manual_seed(0)
    h = torch.randn(n, d, device="cuda", generator=g).bfloat16()
    dhn = (torch.randn(n, d, device="cuda", generator=g) / d ** 0.5).bfloat16()
    wq = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
    wk = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
    bq = (0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16()
    bk = (0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16()

    def fwd():
        return ops.lis_train_fwd(h, wq, bq, wk, bk, k)

    def bwd(o):
        h_new, ps, y, scores, ts, bce = o
        return ops.lis_train_bwd(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, None, 0.7, need_dh=False)

    o = fwd()
    bwd(o)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    iters = 50
    e[0].record()
    for _ in range(iters):
        o = fwd()
    e[1].record()
    for _ in range(iters):
        bwd(o)
    e[2].record()
    torch.cuda.synchronize()
    _native.profile_start()
    o = fwd()
    bwd(o)
    torch.cuda.synchronize()
    prof = _native.profile_stop()
    res[n] = {"k": k, "fwd_us": e[0].elapsed_time(e[1]) / iters * 1e3, "bwd_us": e[1].elapsed_time(e[2]) / iters * 1e3,
              "kernels_us_instrumented": {kk: round(v[0] / v[1] * 1e3, 1) for kk, v in prof.items()}}
print(json.dumps(res, indent=1))
