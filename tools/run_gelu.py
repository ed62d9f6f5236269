#!/usr/bin/env python3
"""python tools/run_gelu.py [n_tokens] [iters]: torch GELU and vsel_gelu_colsum on the merger's hidden activation [n, 5120] (profilers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visionselector_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 147456
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
x = torch.randn(n, 5120, device="cuda", generator=torch.Generator(device="cuda").manual_seed(0)).bfloat16()
for name, f in (("torch_gelu", lambda: torch.nn.functional.gelu(x)), ("gelu_colsum", lambda: ops.gelu_colsum(x, 1))):
    for _ in range(3):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / iters * 1e6:.1f} us")
y, s = ops.gelu_colsum(x, 1)
print("bit-identical to torch:", bool(torch.equal(y, torch.nn.functional.gelu(x))))
