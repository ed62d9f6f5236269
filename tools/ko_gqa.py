import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from visionselector_amd import _native
_native.LIB_PATH = sys.argv[1]
from visionselector_amd import ops
N = _native
n_seq, L = 32, 524
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(n_seq * L, 28, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(n_seq * L, 4, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(n_seq * L, 4, 128, device="cuda", generator=g).bfloat16()
cu = torch.arange(0, n_seq * L + 1, L, device="cuda", dtype=torch.int32)
with N.debug_knob(attn_gqa=1):
    for _ in range(10): ops.varlen_attn(q, k, v, cu, L)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(50): ops.varlen_attn(q, k, v, cu, L)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
print(os.path.basename(sys.argv[1]), f"{best:.1f} us")
