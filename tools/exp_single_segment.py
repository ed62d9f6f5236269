"""Per-kernel time of vsel_lis_select for ONE segment (the reference's joint selection over all visual tokens of a prompt)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N  # noqa: E402
from visionselector_amd import ops  # noqa: E402

g = torch.Generator(device="cuda").manual_seed(0)
for d, hd, n in ((3584, 1792, 2304), (4096, 2048, 5832), (3584, 1792, 16384), (3584, 1792, 65536), (3584, 1792, 147456)):
    k = int(n * 0.2)
    h = torch.randn(n, d, device="cuda", generator=g).bfloat16()
    wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
    for _ in range(3):
        ops.lis_select(h, wq, bq, wk, bk, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.lis_select(h, wq, bq, wk, bk, k)
    e1.record()
    torch.cuda.synchronize()
    N.profile_start()
    for _ in range(10):
        ops.lis_select(h, wq, bq, wk, bk, k)
    prof = N.profile_stop()
    print(json.dumps({"n": n, "d": d, "k": k, "total_us": round(e0.elapsed_time(e1) / 20 * 1e3, 1),
                      "kernels_us": {kk: round(v[0] / v[1] * 1e3, 1) for kk, v in prof.items()}}))
