"""One attention-backward configuration, a few launches (for rocprofv3 --kernel-trace / --pmc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops  # noqa: E402

n_seq, L = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
hq, hkv = 28, 4
t = n_seq * L
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
do = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
cu = torch.arange(0, t + 1, L, dtype=torch.int32, device="cuda")
out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
for _ in range(iters):
    ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
torch.cuda.synchronize()
