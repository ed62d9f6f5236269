"""hipGraph capture of lis_select (no sync / no hipMalloc inside the C-ABI): single-image latency, eager vs graph replay."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops
n, d, hd, k = 2304, 3584, 1792, 460
g = torch.Generator(device="cuda").manual_seed(0)
for b in (1, 4, 16):
    h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
    wq = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16(); wk = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
    bq = torch.zeros(hd, device="cuda").bfloat16(); bk = bq.clone()
    ref = ops.lis_select(h, wq, bq, wk, bk, k)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): ops.lis_select(h, wq, bq, wk, bk, k)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        out = ops.lis_select(h, wq, bq, wk, bk, k)
    graph.replay(); torch.cuda.synchronize()
    same = all(torch.equal(a, c) for a, c in zip(ref, out))
    def timeit(f, it=200):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(it): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / it * 1e6
    te = timeit(lambda: ops.lis_select(h, wq, bq, wk, bk, k))
    tg = timeit(graph.replay)
    print(f"B={b}: eager {te:.1f} us/call, graph replay {tg:.1f} us/call, identical={same}")
