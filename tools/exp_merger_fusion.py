"""Section 8f N2 measurement: merger GELU + un-reorder + LIS at Qwen2.5-VL-7B geometry (merger hidden 5120, D 3584).
 (a) torch GELU, then vsel_lis_select_permuted (two sweeps over H)
 (b) vsel_gelu_colsum, one skinny fp32 GEMM, then vsel_lis_select_presummed (one sweep over H)
The merger's two big GEMMs are identical in both and left out."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops  # noqa: E402

c, d, hd = 5120, 3584, 1792
g = torch.Generator(device="cuda").manual_seed(0)
w2 = (0.02 * torch.randn(d, c, device="cuda", generator=g)).bfloat16()
b2 = (0.02 * torch.randn(d, device="cuda", generator=g)).bfloat16()
w2f_t = w2.float().t().contiguous()
wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for n in (2304, 4 * 2304, 8 * 2304, 16 * 2304, 64 * 2304):
    k = int(n * 0.2)
    x = torch.randn(n, c, device="cuda", generator=g).bfloat16()          # output of the merger's first Linear
    h = torch.randn(n, d, device="cuda", generator=g).bfloat16()          # output of its second Linear (window order)
    perm = torch.randperm(n, device="cuda", generator=g)
    p2l = torch.empty_like(perm)
    p2l[perm] = torch.arange(n, device="cuda")

    def two_sweep():
        torch.nn.functional.gelu(x)
        return ops.lis_select_permuted(h, perm, p2l, wq, bq, wk, bk, k)

    def one_sweep():
        _, sums = ops.gelu_colsum(x, 1)
        cs = ops.colsum_linear(sums, w2, b2, n)
        return ops.lis_select_presummed(h, cs, wq, bq, wk, bk, k, logical_to_physical=perm, physical_to_logical=p2l)

    res = {"n_tokens": n, "torch_gelu_us": timeit(lambda: torch.nn.functional.gelu(x)),
           "gelu_colsum_us": timeit(lambda: ops.gelu_colsum(x, 1)),
           "colsum_linear_us": timeit(lambda: ops.colsum_linear(ops.gelu_colsum(x, 1)[1], w2, b2, n)) - timeit(lambda: ops.gelu_colsum(x, 1)),
           "torch_addmm_fp32_us": timeit(lambda: torch.addmm(b2.float() * n, torch.zeros(1, c, device="cuda"), w2.float().t())),
           "lis_two_sweep_us": timeit(lambda: ops.lis_select_permuted(h, perm, p2l, wq, bq, wk, bk, k)),
           "gelu+lis_two_sweep_us": timeit(two_sweep), "gelu+lis_one_sweep_us": timeit(one_sweep)}
    res["saving"] = 1 - res["gelu+lis_one_sweep_us"] / res["gelu+lis_two_sweep_us"]
    print(json.dumps({k2: (round(v, 1) if isinstance(v, float) and v > 1 else round(v, 3) if isinstance(v, float) else v)
                      for k2, v in res.items()}))
