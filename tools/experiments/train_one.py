"""One training micro-batch (n = 2304) repeated: for rocprofv3 --kernel-trace (per-kernel durations and start gaps of the training step)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionselector_amd import ops
d, hd, n = 3584, 1792, int(sys.argv[1]) if len(sys.argv) > 1 else 2304
k = int(n * 0.2)
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(n, d, device="cuda", generator=g).bfloat16()
dhn = (torch.randn(n, d, device="cuda", generator=g) / d ** 0.5).bfloat16()
wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
bucket = torch.zeros(2 * (hd * d + hd), dtype=torch.float32, device="cuda")
views, off = [], 0
for shape in ((hd, d), (hd,), (hd, d), (hd,)):
    cnt = shape[0] * (shape[1] if len(shape) > 1 else 1)
    views.append(bucket[off:off + cnt].view(*shape)); off += cnt
for _ in range(60):
    h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, k)
    ops.lis_train_bwd(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, None, 0.7, need_dh=False, out=views)
torch.cuda.synchronize()
