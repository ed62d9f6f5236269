import sys, os, torch, numpy as np
sys.path.insert(0, "/root/repo")
from visionselector_amd import ops, _native as N
from oracle import inputs as oin
d, hd, n = 3584, 1792, 2304
k = int(n * 0.2)
c = oin.make_case(d, hd, n, 123)
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda().bfloat16()
h, wq, bq, wk, bk = (dev(c[x]) for x in ("h", "wq", "bq", "wk", "bk"))
g = torch.Generator(device="cuda").manual_seed(11)
dhn = (torch.randn(n, d, device="cuda", generator=g) / d ** 0.5).bfloat16()
h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(h, wq, bq, wk, bk, k)
res = {}
for fused in (1, 0):
    with N.debug_knob("train_fused", fused):
        payload, dh = ops.lis_train_bwd_factors(dhn, h, wq, bq, wk, bk, ps, y, scores, ts, None, 0.7, need_dh=True)
        res[fused] = ops.factor_payload_split(payload, hd, d) + (dh,)
for nm, a, b in zip(("a", "gx", "dk", "xsum", "dbq", "dbk", "dh"), res[1], res[0]):
    diff = (a.float() - b.float()).abs()
    print(nm, bool(torch.equal(a, b)), int((diff > 0).sum()), float(diff.max()), float(b.float().abs().max()))
