"""Same-process A/B of the dK / dV walk directions (knob attn_bwd_updown: odd kv heads walk their query tiles upward, pairs queued in couples per
XCD) at 7B heads: dK / dV kernel us per setting, two alternating rounds, and the relative difference of the gradients (dQ 0, dK / dV a bf16 rounding).
    python tools/exp_dkdv_updown.py   ->  profiles/r05_dkdv_tile_order.txt"""
import sys, os, json, torch
sys.path.insert(0, os.getcwd())
from visionselector_amd import _native as N, ops
def run(n_seq, L, hq=28, hkv=4, iters=20):
    t = n_seq * L
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, t + 1, L, dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
    res = {}
    for rnd in range(2):
        for ud in (0, 1):
            with N.debug_knob(attn_bwd_updown=ud):
                for _ in range(3): ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                torch.cuda.synchronize()
                N.profile_start()
                for _ in range(iters): g_ = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                prof = N.profile_stop()
                us = {n: 1e3 * ms / c for n, (ms, c) in prof.items()}
                res.setdefault(ud, []).append(round(sum(x for n, x in us.items() if "dkdv" in n), 1))
                if rnd == 0: res[("g", ud)] = g_
    d = [float((a.float() - b.float()).abs().max() / b.float().abs().max()) for a, b in zip(res[("g", 1)], res[("g", 0)])]
    print(json.dumps({"n_seq": n_seq, "L": L, "hq": hq, "hkv": hkv, "dkdv_us_updown0": res[0], "dkdv_us_updown1": res[1], "rel_diff_dq_dk_dv": d}))
for s in [(16, 3072), (16, 3584), (16, 3840), (16, 4096), (16, 4608), (12, 5120), (8, 6144), (8, 8192), (32, 4096), (8, 4096)]:
    run(*s)
