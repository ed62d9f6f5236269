"""Same-process A/B of the group-shared short-sequence forward (csrc/attn_fwd_gqa.hip, knob attn_gqa) against the per-head forms.
    python tools/exp_attn_gqa.py  ->  one line per shape: us and TFLOP/s of both, bit-equality"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from visionselector_amd import ops, _native as N


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


def main():
    rng = np.random.default_rng(5)
    ragged64 = [int(x) for x in rng.integers(131, 948, size=64)]
    ragged8 = ragged64[:8]
    shapes = [("1x524", [524], 28, 4), ("2x524", [524] * 2, 28, 4), ("4x524", [524] * 4, 28, 4), ("8x524", [524] * 8, 28, 4),
              ("16x524", [524] * 16, 28, 4), ("32x524", [524] * 32, 28, 4), ("64x524", [524] * 64, 28, 4), ("32x294", [294] * 32, 28, 4),
              ("32x1216", [1216] * 32, 28, 4), ("16x1100", [1100] * 16, 28, 4), ("8x2000", [2000] * 8, 28, 4), ("4x2368", [2368] * 4, 28, 4),
              ("ragged8", ragged8, 28, 4), ("ragged64", ragged64, 28, 4),
              ("16x4096", [4096] * 16, 28, 4), ("4x8192", [8192] * 4, 28, 4), ("16x2368", [2368] * 16, 28, 4),
              ("ov 32x1230", [1230] * 32, 32, 8), ("ov 8x1230", [1230] * 8, 32, 8), ("3b 32x524", [524] * 32, 16, 2)]
    for tag, lens, hq, hkv in shapes:
        total = sum(lens)
        g = torch.Generator(device="cuda").manual_seed(3)
        q = torch.randn(total, hq, 128, device="cuda", generator=g).bfloat16()
        k = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
        v = torch.randn(total, hkv, 128, device="cuda", generator=g).bfloat16()
        cu = torch.from_numpy(np.concatenate(([0], np.cumsum(lens))).astype(np.int32)).cuda()
        L = max(lens)
        fl = sum(4.0 * l * l * hq * 128 / 2 for l in lens)
        res = {}
        outs = {}
        for name, knobs in (("default", {}), ("per_head", dict(attn_gqa=0)), ("gqa", dict(attn_gqa=1, attn_gqa_form=0)), ("gqa64", dict(attn_gqa=1, attn_gqa_form=1))):
            with N.debug_knob(**knobs):
                outs[name] = ops.varlen_attn(q, k, v, cu, L)
                res[name] = timed(lambda: ops.varlen_attn(q, k, v, cu, L))
        eq64 = torch.equal(outs["per_head"], outs["gqa64"])
        eq = torch.equal(outs["per_head"], outs["gqa"])
        md = (outs["per_head"].float() - outs["gqa"].float()).abs().max().item()
        print(f"{tag:12s} default {res['default']:8.1f} us  per_head {res['per_head']:8.1f} us {fl / res['per_head'] / 1e6:7.0f} TF   "
              f"gqa {res['gqa']:8.1f} us {fl / res['gqa'] / 1e6:7.0f} TF   x{res['per_head'] / res['gqa']:.2f}  equal={eq} | gqa64 {res['gqa64']:8.1f} us {fl / res['gqa64'] / 1e6:7.0f} TF x{res['per_head'] / res['gqa64']:.2f} equal={eq64}", flush=True)


if __name__ == "__main__":
    main()
