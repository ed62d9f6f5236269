#!/usr/bin/env python3
"""dK / dV pass, item forms by shape: the group's q heads inside the item (attn_bwd_split = 0), one q head per item (1), the group in
two parts (2, dkdv64 only) and the library's rule (-1).  Per-kernel times (HIP events of the library's profiler; the partial forms'
reduce kernel counted with them), same process, alternating; dK / dV of the partial forms against the unsplit form (max |diff| relative
to the tensor's max: fp32 association differs, bf16 outputs agree to a rounding)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N, ops

def ragged(n, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, (n,), generator=g).tolist()

cases = [(f"{n}x{l}", [l] * n) for n, l in ((2, 1100), (3, 1100), (4, 1100), (5, 1100), (6, 1100), (7, 1100), (9, 1100), (2, 2000), (3, 2000), (4, 2000), (1, 2368), (2, 2368), (3, 2368),
                                            (1, 3000), (1, 4096), (2, 4096), (1, 8192))]
if len(sys.argv) > 1 and sys.argv[1] == "short":       # below 1024 tokens: the 8-wave kernel
    cases = [(f"{n}x{l}", [l] * n) for n, l in ((4, 524), (6, 524), (8, 524), (10, 524), (12, 524), (16, 524), (20, 524), (24, 524), (32, 524), (12, 300), (16, 300), (24, 300), (32, 300), (48, 300),
                                                (4, 800), (6, 800), (8, 800), (12, 800), (16, 800))]
    cases += [("c5_8", ragged(8, 131, 947, 4)), ("c5_12", ragged(12, 131, 947, 12)), ("c5_16", ragged(16, 131, 947, 16)), ("c5_24", ragged(24, 131, 947, 24))]
if len(sys.argv) > 1 and sys.argv[1] == "ragged":
    cases = [(f"c5_{n}", ragged(n, 131, 947, n)) for n in (6, 8, 12, 16, 24, 32)] + [(f"mix2k_{n}", ragged(n, 512, 2040, n)) for n in (4, 6, 9, 14)]
    cases += [(f"mix4k_{n}", ragged(n, 1024, 4096, n)) for n in (2, 3, 5, 8)] + [("two_classes", [300] * 30 + [1800] * 6), ("long_short", [4000] + [200] * 40)]
if len(sys.argv) > 1 and sys.argv[1] == "wide":
    cases = [(f"{n}x{l}", [l] * n) for n, l in ((6, 1100), (8, 1100), (10, 1100), (12, 1100), (14, 1100), (6, 1500), (3, 2368), (4, 2368), (5, 2368), (6, 2368),
                                                (2, 4096), (3, 4096), (4, 4096), (1, 8192), (2, 8192), (16, 2368))]
    cases += [("mix4k_5", ragged(5, 1024, 4096, 5)), ("mix2k_9", ragged(9, 512, 2040, 9))]
DKDV = ("attn_bwd_dkdv_kernel", "attn_bwd_dkdv64_kernel", "attn_bwd_group_sum_kernel")
MODES = (0, 1, 2, 3, 4, -1)
for name, lens in cases:
    g = torch.Generator(device="cuda").manual_seed(7)
    T, L = sum(lens), max(lens)
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
    res, grads = {m: [] for m in MODES}, {}
    for rnd in range(2):
        for m in MODES:
            with N.debug_knob(attn_bwd_split=m):
                for _ in range(4):
                    gr = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                N.profile_start()
                for _ in range(6):
                    ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                prof = N.profile_stop()
            grads[m] = gr
            res[m].append(round(sum(prof[n][0] / prof[n][1] * 1e3 for n in prof if any(n.startswith(d) for d in DKDV)), 1))
    r = {m: min(x) for m, x in res.items()}
    rel = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max())  # noqa: E731
    items = -(-L // 128) * 4 * len(lens)
    print(json.dumps({"case": name, "tokens": T, "unsplit_items": items, "us_group": r[0], "us_per_head": r[1], "us_two_parts": r[2], "us_three_parts": r[3], "us_four_parts": r[4], "us_rule": r[-1],
                      "rule_vs_best_old": round(min(r[0], r[1]) / r[-1], 3), "dq_equal": bool(torch.equal(grads[0][0], grads[2][0])),
                      "dk_rel_parts": rel(grads[2][1], grads[0][1]), "dv_rel_parts": rel(grads[2][2], grads[0][2]),
                      "dk_rel_heads": rel(grads[1][1], grads[0][1])}), flush=True)
