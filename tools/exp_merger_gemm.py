"""What a merger GEMM with a column-sum epilogue would have to match: torch (hipBLASLt) bf16 Linear(5120 -> 3584) at the merger's
token counts, and the cost of the two alternatives that give the LIS its column sums without touching the GEMM."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lin = torch.nn.Linear(5120, 3584, bias=True, device="cuda", dtype=torch.bfloat16)


def timeit(fn, iters=10):
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


for n in (2304, 36864, 147456):
    x = torch.randn(n, 5120, device="cuda").bfloat16()
    us = timeit(lambda: lin(x))
    h = lin(x)
    us_sum = timeit(lambda: h.float().sum(0))          # what a separate column-sum pass over H costs in torch
    print(json.dumps({"tokens": n, "linear_us": round(us, 1), "TFLOPs": round(2.0 * n * 5120 * 3584 / us / 1e6, 1),
                      "torch_colsum_of_H_us": round(us_sum, 1), "sweep1_bytes_MB": round(n * 3584 * 2 / 1e6, 1)}))
