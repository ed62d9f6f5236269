import sys, os, time, torch
sys.path.insert(0, "/root/repo")
from visionselector_amd import ops
n, d, hd, k = 2304, 3584, 1792, 460
g = torch.Generator(device="cuda").manual_seed(0)
wq = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16(); wk = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
bq = torch.zeros(hd, device="cuda").bfloat16(); bk = bq.clone()
res = []
for b in (1, 4, 8, 16, 32, 64):
    h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
    for _ in range(5): ops.lis_select(h, wq, bq, wk, bk, k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it = 200
    for _ in range(it): ops.lis_select(h, wq, bq, wk, bk, k)
    torch.cuda.synchronize(); res.append((b, round((time.perf_counter() - t0) / it * 1e6, 1)))
print(res)
