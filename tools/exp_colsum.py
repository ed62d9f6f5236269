import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops, _native
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n, d, hd, k = 2304, 3584, 1792, 460
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
wq = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16(); wk = wq.clone()
bq = torch.zeros(hd, device="cuda").bfloat16(); bk = bq.clone()
for mode in ("scores_only", "select"):
    f = (lambda: ops.lis_scores(h, wq, bq, wk, bk)) if mode == "scores_only" else (lambda: ops.lis_select(h, wq, bq, wk, bk, k))
    for _ in range(3): f()
    torch.cuda.synchronize()
    _native.profile_start()
    for _ in range(10): f()
    torch.cuda.synchronize()
    p = _native.profile_stop()
    print(b, mode, {kk: round(v[0] / v[1] * 1e3, 1) for kk, v in p.items() if "colsum_" in kk or "score" in kk or "gather" in kk})
