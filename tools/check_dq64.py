#!/usr/bin/env python3
"""The 64-rows-per-wave dQ pass (csrc/attn_bwd_dq64.hip, knob attn_bwd_dq64) against attn_bwd_dq_kernel: dQ must be bit-identical, and
so must dK / dV (they consume the D and lse2 the dQ pass leaves in the workspace); then per-kernel times on the large shapes.

    python tools/check_dq64.py [--no-bench] [--quick]"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N, ops
if os.environ.get("VSEL_CHECK_LIB"):          # a variant library of tools/ab_fwd64.py instead of the shipped one
    N.LIB_PATH = os.environ["VSEL_CHECK_LIB"]


def grads(q, k, v, do, cu, L, causal, dq64):
    with N.debug_knob(attn_rows64=0, attn_split=0):
        out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L, causal=causal)
    with N.debug_knob(attn_bwd_dq64=dq64):
        N.profile_start()
        res = ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L, causal=causal)
        prof = N.profile_stop()
    assert ("attn_bwd_dq64_kernel" in prof) == bool(dq64), prof
    return res, prof


def case(lens, hq, hkv, causal, seed=0, scale_q=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    T = sum(lens)
    q = (torch.randn(T, hq, 128, device="cuda", generator=g) * scale_q).bfloat16()
    k = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, hkv, 128, device="cuda", generator=g).bfloat16()
    do = torch.randn(T, hq, 128, device="cuda", generator=g).bfloat16()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    L = max(lens)
    (a, _), (b, _) = grads(q, k, v, do, cu, L, causal, 0), grads(q, k, v, do, cu, L, causal, 1)
    torch.cuda.synchronize()
    rec = {"lens": lens if len(lens) <= 4 else f"{len(lens)} x ...", "hq": hq, "hkv": hkv, "causal": causal}
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
    cases = [([64], 4, 4, True), ([256], 4, 4, True), ([300], 4, 2, True), ([1000], 8, 2, True), ([1000], 8, 2, False),
             ([37, 700, 256, 129], 28, 4, True), ([524] * 4, 28, 4, True), ([2368], 28, 4, True), ([2368], 4, 4, False)]
    if "--quick" in sys.argv:
        cases = cases[:4]
    for lens, hq, hkv, causal in cases:
        ok &= case(lens, hq, hkv, causal)
    ok &= case([1500], 8, 2, True, seed=3, scale_q=4.0)
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
            for dq64 in (0, 1):
                with N.debug_knob(attn_bwd_dq64=dq64):
                    for _ in range(8):
                        ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                    N.profile_start()
                    for _ in range(5):
                        ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
                    prof = N.profile_stop()
                name = "attn_bwd_dq64_kernel" if dq64 else "attn_bwd_dq_kernel"
                res.setdefault(name, []).append(round(prof[name][0] / prof[name][1] * 1e3, 1))
        print(json.dumps({"n_seq": nseq, "L": L, "dq_kernel_us": res}), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
