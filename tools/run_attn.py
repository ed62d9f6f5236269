import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2368
g = torch.Generator(device="cuda").manual_seed(7)
q = torch.randn(L, 28, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(L, 4, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(L, 4, 128, device="cuda", generator=g).bfloat16()
cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
for _ in range(5):
    ops.varlen_attn(q, k, v, cu, L)
torch.cuda.synchronize()
