import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops
for L in (64, 128, 256, 384, 524, 1024):
    g = torch.Generator(device="cuda").manual_seed(7)
    q = torch.randn(L, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(L, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(L, 4, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
    for causal in (True, False):
        for _ in range(5): ops.varlen_attn(q, k, v, cu, L, causal=causal)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.varlen_attn(q, k, v, cu, L, causal=causal)
        e1.record(); torch.cuda.synchronize()
        print(f"L={L} causal={causal}: {e0.elapsed_time(e1)/50*1e3:.1f} us  ({(L+63)//64} kv tiles max)")
