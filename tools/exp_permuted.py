import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops
b, n, d, hd, k = 128, 2304, 3584, 1792, 460
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
wq = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16(); wk = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
bq = torch.zeros(hd, device="cuda").bfloat16(); bk = bq.clone()
cg = torch.Generator().manual_seed(1)
p2l = torch.cat([torch.randperm(n, generator=cg) + i * n for i in range(b)]).cuda()
l2p = torch.argsort(p2l)
def t(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
flat = h.view(b * n, d)
ref = t(lambda: ops.lis_select(flat[l2p].view(b, n, d), wq, bq, wk, bk, k))
unre = t(lambda: flat[l2p])
sel = t(lambda: ops.lis_select(h, wq, bq, wk, bk, k))
per = t(lambda: ops.lis_select_permuted(h, l2p, p2l, wq, bq, wk, bk, k))
print(f"B={b}: reference order of ops (torch un-reorder gather + lis_select) {ref:.0f} us  [un-reorder alone {unre:.0f} us, lis_select alone {sel:.0f} us];  lis_select_permuted {per:.0f} us")
