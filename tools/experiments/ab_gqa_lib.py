"""us per call of the default attention forward for a few packed short shapes, for a given build of the library (--lib)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from visionselector_amd import _native as N
if "--lib" in sys.argv:
    N.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
import numpy as np, torch
from visionselector_amd import ops
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
rng = np.random.default_rng(5)
ragged64 = [int(x) for x in rng.integers(131, 948, size=64)]
out = {}
for tag, lens, hq, hkv in (("8x524", [524] * 8, 28, 4), ("32x524", [524] * 32, 28, 4), ("32x294", [294] * 32, 28, 4), ("ragged64", ragged64, 28, 4),
                           ("3b 32x524", [524] * 32, 16, 2), ("ov 32x1230", [1230] * 32, 32, 8)):
    total = sum(lens)
    g = torch.Generator(device="cuda").manual_seed(3)
    q = torch.randn(total, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32, device="cuda")
    with N.debug_knob(attn_gqa=1, attn_gqa_form=0):
        us = timed(lambda: ops.varlen_attn(q, k, v, cu, max(lens)))
        o = ops.varlen_attn(q, k, v, cu, max(lens))
    fl = sum(4.0 * L * L * hq * 128 / 2 for L in lens)
    out[tag] = (round(us, 1), round(fl / us / 1e6), float(o.float().abs().sum()))
print(json.dumps(out))
