"""Level-major vs pair-major deal of the group-shared forward's item list (knob attn_pair_major), same process, bit equality."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from visionselector_amd import _native as N, ops
def timed(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best
forms = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0").split(",")]
for form in forms:
  for tag, lens, hq, hkv in (("4x524", [524] * 4, 28, 4), ("8x524", [524] * 8, 28, 4), ("16x524", [524] * 16, 28, 4), ("32x524", [524] * 32, 28, 4), ("64x524", [524] * 64, 28, 4),
                           ("32x294", [294] * 32, 28, 4), ("16x1100", [1100] * 16, 28, 4), ("3b 32x524", [524] * 32, 16, 2), ("ov 32x1230", [1230] * 32, 32, 8), ("ov 8x1230", [1230] * 8, 32, 8)):
    total = sum(lens)
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(total, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32, device="cuda")
    res = {}
    outs = {}
    for rnd in range(2):
        for pm in (0, 1):
            with N.debug_knob(attn_gqa=1, attn_gqa_form=form, attn_pair_major=pm):
                us = timed(lambda: ops.varlen_attn(q, k, v, cu, max(lens)))
                outs[pm] = ops.varlen_attn(q, k, v, cu, max(lens))
            res.setdefault(pm, []).append(round(us, 1))
    fl = sum(4.0 * L * L * hq * 128 / 2 for L in lens)
    print(json.dumps({"form": form, "shape": tag, "level_major_us": res[0], "pair_major_us": res[1], "pair_major_tflops": round(fl / min(res[1]) / 1e6),
                      "equal": bool(torch.equal(outs[0], outs[1]))}), flush=True)
