"""What PyTorch-ROCm's own fused attention (scaled_dot_product_attention: AOTriton / CK flash kernels) does on the same box at
the bench shapes -- a yardstick for the vsel forward / backward kernels, not a dependency.  7B heads (28 q / 4 kv, d 128), causal."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N  # noqa: E402
from visionselector_amd import ops  # noqa: E402


def ev(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for n_seq, L in ((16, 4096), (4, 8192), (16, 2368)):
    hq, hkv, d = 28, 4, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(n_seq, hq, L, d, device="cuda", generator=g).bfloat16().requires_grad_(True)
    k = torch.randn(n_seq, hkv, L, d, device="cuda", generator=g).bfloat16().requires_grad_(True)
    v = torch.randn(n_seq, hkv, L, d, device="cuda", generator=g).bfloat16().requires_grad_(True)
    do = torch.randn(n_seq, hq, L, d, device="cuda", generator=g).bfloat16()
    flops_f = 4 * L * L * hq * d / 2 * n_seq
    res = {"n_seq": n_seq, "L": L}
    for name, kw in (("sdpa_gqa", dict(enable_gqa=True)),):
        try:
            fwd = lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True, **kw)  # noqa: E731
            t_f = ev(lambda: fwd().detach())
            o = fwd()
            t_b = ev(lambda: torch.autograd.grad(o, (q, k, v), do, retain_graph=True))
            res[name] = {"fwd_us": round(t_f, 1), "fwd_TF": round(flops_f / t_f / 1e6, 1), "bwd_us": round(t_b, 1),
                         "bwd_TF_alg": round(2.5 * flops_f / t_b / 1e6, 1)}
        except Exception as e:  # noqa: BLE001
            res[name] = {"error": str(e)[:200]}
    # ours on the packed layout
    t = n_seq * L
    qp = q.detach().transpose(1, 2).reshape(t, hq, d).contiguous()
    kp = k.detach().transpose(1, 2).reshape(t, hkv, d).contiguous()
    vp = v.detach().transpose(1, 2).reshape(t, hkv, d).contiguous()
    dop = do.transpose(1, 2).reshape(t, hq, d).contiguous()
    cu = torch.arange(0, t + 1, L, dtype=torch.int32, device="cuda")
    t_f = ev(lambda: ops.varlen_attn(qp, kp, vp, cu, L))
    out, lse = ops.varlen_attn_fwd_lse(qp, kp, vp, cu, L)
    t_b = ev(lambda: ops.varlen_attn_bwd(dop, qp, kp, vp, out, lse, cu, L))
    res["vsel"] = {"fwd_us": round(t_f, 1), "fwd_TF": round(flops_f / t_f / 1e6, 1), "bwd_us": round(t_b, 1),
                   "bwd_TF_alg": round(2.5 * flops_f / t_b / 1e6, 1)}
    print(json.dumps(res), flush=True)
