"""Same-process A/B of the group-shared attention forms with their item lists DEALT (knob attn_static = 1: mirrored rounds, no counter) and
DRAWN from the queue behind two dealt rounds (attn_static = 0), against the per-head form and the library default; bit-equality checked.
    python tools/exp_attn_deal.py     ->  profiles/r05_gqa_ab.txt, second table"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from visionselector_amd import ops, _native as N
from tools.exp_attn_gqa import timed
rng = np.random.default_rng(5)
ragged64 = [int(x) for x in rng.integers(131, 948, size=64)]
shapes = [("4x524", [524] * 4, 28, 4), ("5x524", [524] * 5, 28, 4), ("6x524", [524] * 6, 28, 4), ("7x524", [524] * 7, 28, 4), ("8x524", [524] * 8, 28, 4), ("32x524", [524] * 32, 28, 4), ("64x524", [524] * 64, 28, 4),
          ("32x294", [294] * 32, 28, 4), ("16x1100", [1100] * 16, 28, 4), ("32x1216", [1216] * 32, 28, 4), ("ragged8", ragged64[:8], 28, 4), ("ragged64", ragged64, 28, 4),
          ("ov 8x1230", [1230] * 8, 32, 8), ("ov 32x1230", [1230] * 32, 32, 8), ("3b 32x524", [524] * 32, 16, 2)]
for tag, lens, hq, hkv in shapes:
    total = sum(lens)
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(total, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
    cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
    L = max(lens)
    fl = sum(4.0 * l * l * hq * 128 / 2 for l in lens)
    res, outs = {}, {}
    for name, knobs in (("per_head", dict(attn_gqa=0)), ("gqa_queue", dict(attn_gqa=1, attn_gqa_form=0, attn_static=0)), ("gqa_deal", dict(attn_gqa=1, attn_gqa_form=0, attn_static=1)),
                        ("gqa64_queue", dict(attn_gqa=1, attn_gqa_form=1, attn_static=0)), ("gqa64_deal", dict(attn_gqa=1, attn_gqa_form=1, attn_static=1)), ("default", {})):
        with N.debug_knob(**knobs):
            outs[name] = ops.varlen_attn(q, k, v, cu, L)
            res[name] = timed(lambda: ops.varlen_attn(q, k, v, cu, L))
    eq = torch.equal(outs["per_head"], outs["gqa_deal"]) and torch.equal(outs["per_head"], outs["gqa_queue"])
    print(f"{tag:12s} " + "  ".join(f"{n} {res[n]:7.1f} us {fl / res[n] / 1e6:5.0f} TF" for n in res) + f"  equal={eq}")
