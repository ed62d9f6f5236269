"""Per-kernel times of one vsel_lis_select configuration for a given build of the library (same-box A/B of variants):
   python tools/ab_lis.py [--lib path/to/libvsel.so] [--images 128] [--iters 30]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N  # noqa: E402
if "--lib" in sys.argv:
    N.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from visionselector_amd import ops  # noqa: E402


def arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default


b, iters = arg("--images", 128), arg("--iters", 30)
n, d, hd, k = 2304, 3584, 1792, 460
g = torch.Generator(device="cuda").manual_seed(1234)
h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
for _ in range(5):
    ops.lis_select(h, wq, bq, wk, bk, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.lis_select(h, wq, bq, wk, bk, k)
e1.record()
torch.cuda.synchronize()
N.profile_start()
for _ in range(iters):
    ops.lis_select(h, wq, bq, wk, bk, k)
prof = N.profile_stop()
print(json.dumps({"lib": os.path.basename(N.LIB_PATH or "libvsel.so"), "images": b, "us_per_step": round(e0.elapsed_time(e1) / iters * 1e3, 1),
                  "kernels_us": {kn: round(ms / c * 1e3, 1) for kn, (ms, c) in prof.items()}}))
