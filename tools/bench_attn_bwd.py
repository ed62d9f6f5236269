"""Time the var-len attention backward (7B geometry: 28 q heads, 4 kv heads, d 128) next to the forward.
FLOPs: forward 4*L^2*h*d/2 (causal), backward 2.5x that (5 contractions vs 2); the kernels recompute S and dP in both the
dQ and the dK/dV pass (7 contractions), the reported TFLOP/s uses the ALGORITHMIC 2.5x."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N  # noqa: E402
if "--lib" in sys.argv:            # another build of the library (same-box A/B runs: tools/ab_attn.sh)
    N.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from visionselector_amd import ops  # noqa: E402


def run(n_seq, L, hq=28, hkv=4, iters=20):
    t = n_seq * L
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, t + 1, L, dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
    for _ in range(3):
        ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
    torch.cuda.synchronize()
    N.profile_start()
    for _ in range(iters):
        ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
    prof = N.profile_stop()
    us = {n: 1e3 * ms / calls for n, (ms, calls) in prof.items()}
    total_us = sum(us.values())
    flops = 2.5 * 4 * L * L * hq * 128 / 2 * n_seq
    print(json.dumps({"n_seq": n_seq, "L": L, "bwd_us": round(total_us, 1), "TFLOPs_alg": round(flops / total_us / 1e6, 1),
                      "kernels_us": {n: round(x, 1) for n, x in us.items()}}))


if __name__ == "__main__":
    shapes = [(1, 524), (1, 2368), (8, 524), (4, 2368), (16, 2368), (16, 4096), (4, 8192), (16, 1100)]
    if "--big" in sys.argv:
        shapes = [(16, 4096), (4, 8192), (16, 1100)]
    if "--shape" in sys.argv:          # --shape N_SEQ L
        i = sys.argv.index("--shape")
        shapes = [(int(sys.argv[i + 1]), int(sys.argv[i + 2]))]
    for n_seq, L in shapes:
        run(n_seq, L)
