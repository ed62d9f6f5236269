#!/usr/bin/env python3
"""Bare loop of ops.lis_select at Qwen2.5-VL-7B geometry for profilers:  python tools/run_lis.py B ITERS [graph|-] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visionselector_amd import ops

b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
use_graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
n, d, hd, k = 2304, 3584, 1792, (int(sys.argv[4]) if len(sys.argv) > 4 else 460)
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
wq = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
wk = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
bq = (0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16()
bk = (0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16()
f = lambda: ops.lis_select(h, wq, bq, wk, bk, k)  # noqa: E731
for _ in range(5):
    f()
torch.cuda.synchronize()
if use_graph:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=s):
        f()
    f = gr.replay
    f()
    torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    f()
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / iters * 1e6
print(f"B={b} k={k} iters={iters} graph={use_graph}: {us:.1f} us/call, {b * n / us:.1f} M tokens/s")
