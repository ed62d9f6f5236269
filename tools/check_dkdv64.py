#!/usr/bin/env python3
"""The one-wave-per-SIMD dK / dV pass (csrc/attn_bwd_dkdv64.hip, knob attn_bwd_dkdv64) against the four-wave attn_bwd_dkdv_kernel<false>
(knob attn_bwd_waves = 4, the same work split and summation order): dK / dV must be bit-identical; then per-kernel times.

    python tools/check_dkdv64.py [--no-bench] [--quick]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N, ops
if os.environ.get("VSEL_CHECK_LIB"):          # a variant library of tools/ab_fwd64.py instead of the shipped one
    N.LIB_PATH = os.environ["VSEL_CHECK_LIB"]


def grads(q, k, v, do, out, lse, cu, L, causal, new, split=0):
    with N.debug_knob(attn_bwd_dkdv64=new, attn_bwd_waves=4, attn_bwd_split=split):
        N.profile_start()
        res = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L, causal=causal)
        prof = N.profile_stop()
    assert ("attn_bwd_dkdv64_kernel" in prof) == bool(new), prof
    return res


def case(lens, hq, hkv, causal, seed=0, split=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    T = sum(lens)
    q = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    L = max(lens)
    out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L, causal=causal)
    a, b = grads(q, k, v, do, out, lse, cu, L, causal, 0, split), grads(q, k, v, do, out, lse, cu, L, causal, 1, split)
    torch.cuda.synchronize()
    rec = {"lens": lens if len(lens) <= 4 else f"{len(lens)} x ...", "hq": hq, "hkv": hkv, "causal": causal, "split": split}
    ok = True
    for name, x, y in zip(("dq", "dk", "dv"), a, b):
        same = torch.equal(x.view(torch.int16), y.view(torch.int16))
        rec[name + "_identical"] = same
        if not same:
            rec[name + "_mismatches"] = int((x.view(torch.int16) != y.view(torch.int16)).sum())
            rec[name + "_max_abs_diff"] = float((x.float() - y.float()).abs().max())
            rec[name + "_finite"] = bool(torch.isfinite(y.float()).all())
        ok &= same
    print(json.dumps(rec), flush=True)
    return ok


def main():
    ok = True
    cases = [([64], 4, 4, True), ([128], 4, 4, True), ([256], 4, 2, True), ([300], 4, 2, True), ([1000], 8, 2, True), ([1000], 8, 2, False),
             ([37, 700, 256, 129], 28, 4, True), ([524] * 4, 28, 4, True), ([2368], 28, 4, True), ([2368], 4, 4, False)]
    if "--quick" in sys.argv:
        cases = cases[:5]
    for lens, hq, hkv, causal in cases:
        ok &= case(lens, hq, hkv, causal)
        if hq != hkv:                  # the per-q-head item form (fp32 partials + attn_bwd_group_sum_kernel), both kernels forced into it
            ok &= case(lens, hq, hkv, causal, split=1)
    print(json.dumps({"all_bit_identical": bool(ok)}), flush=True)
    if "--no-bench" in sys.argv:
        return 0 if ok else 1
    for nseq, L in [(16, 4096), (4, 8192), (16, 2368)]:
        g = torch.Generator(device="cuda").manual_seed(7)
        T = nseq * L
        q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
        k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
        v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
        do = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
        cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
        out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
        res = {}
        for rnd in range(2):
            for new in (0, 1):
                with N.debug_knob(attn_bwd_dkdv64=new):
                    for _ in range(8):
                        ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                    N.profile_start()
                    for _ in range(5):
                        ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                    prof = N.profile_stop()
                name = "attn_bwd_dkdv64_kernel" if new else "attn_bwd_dkdv_kernel"
                res.setdefault(name, []).append(round(prof[name][0] / prof[name][1] * 1e3, 1))
        print(json.dumps({"n_seq": nseq, "L": L, "dkdv_kernel_us": res}), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
