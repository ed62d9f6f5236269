"""python tools/run_attn_batch.py N_SEQ L [unused] [iters]: the attention forward on a packed batch (profilers)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visionselector_amd import ops, _native
nseq, L = int(sys.argv[1]), int(sys.argv[2])
pp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
g = torch.Generator(device="cuda").manual_seed(7)
T = nseq * L
q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
for _ in range(2):
    ops.varlen_attn(q, k, v, cu, L)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters):
    ops.varlen_attn(q, k, v, cu, L)
torch.cuda.synchronize()
us = (time.perf_counter() - t0) / iters * 1e6
print(f"{nseq}x{L}: {us:.1f} us, {4.0 * L * L * 28 * 128 / 2 * nseq / us / 1e6:.1f} TFLOP/s")
