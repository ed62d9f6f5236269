#!/usr/bin/env python3
"""In-kernel timeline of the group-shared short-sequence forward (csrc/attn_fwd_gqa.hip): s_memtime stamps of waves 0 and 5 of the first
workgroups.  Build first:  tools/build_variant.sh gqatrace attn_fwd_gqa.hip -DVSEL_GQA_TRACE
    python tools/trace_gqa.py [n_seq] [L]
tags: 1 item start (Q fragments read) | 2 round start | 3 loads issued | 4 tile body done | 5 barrier passed | 6 epilogue done | 7 next item ready"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "variants", "libvsel_gqatrace.so")


def main():
    import numpy as np
    import torch
    from visionselector_amd import _native
    _native.LIB_PATH = LIB
    from visionselector_amd import ops
    n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 524
    hq, hkv, d = 28, 4, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(n_seq * L, hq, d, device="cuda", generator=g).bfloat16()
    k = torch.randn(n_seq * L, hkv, d, device="cuda", generator=g).bfloat16()
    v = torch.randn(n_seq * L, hkv, d, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, n_seq * L + 1, L, device="cuda", dtype=torch.int32)
    lib = _native.lib()
    lib.vsel_debug_read_gqa_trace.argtypes = [C.c_void_p, C.c_int]
    lib.vsel_debug_read_gqa_trace.restype = C.c_int
    buf = np.zeros((8, 2, 1024), dtype=np.uint64)
    with _native.debug_knob(attn_gqa=1):
        for _ in range(5):
            ops.varlen_attn(q, k, v, cu, L)
        torch.cuda.synchronize()
        lib.vsel_debug_read_gqa_trace(buf.ctypes.data, 1)
        ops.varlen_attn(q, k, v, cu, L)
        torch.cuda.synchronize()
        assert lib.vsel_debug_read_gqa_trace(buf.ctypes.data, 1) == 0
    names = {1: "item", 2: "round", 3: "issued", 4: "body", 5: "barrier", 6: "epilogue", 7: "next"}
    for b in (0, 3):
        for w in (0, 1):
            ev = [(int(x) >> 56, int(x) & ((1 << 56) - 1)) for x in buf[b, w] if x]
            if not ev:
                continue
            t0 = ev[0][1]
            print(f"--- workgroup {b}, wave {'0' if w == 0 else '5'}: {len(ev)} stamps, {ev[-1][1] - t0} cycles in all")
            prev = t0
            line = []
            for tag, t in ev:
                if tag == 1 and line:
                    print("   " + " ".join(line))
                    line = []
                line.append(f"{names[tag]}+{t - prev}")
                prev = t
            print("   " + " ".join(line))
    # aggregate over the 8 workgroups, wave 0: cycles per phase
    agg = {}
    for b in range(8):
        ev = [(int(x) >> 56, int(x) & ((1 << 56) - 1)) for x in buf[b, 0] if x]
        for (tg0, t0_), (tg1, t1_) in zip(ev, ev[1:]):
            agg.setdefault((tg0, tg1), []).append(t1_ - t0_)
    print("phase (from -> to): count, median, mean cycles, share of the total")
    tot = sum(sum(v_) for v_ in agg.values())
    for (a, b_), v_ in sorted(agg.items()):
        print(f"  {names[a]:8s} -> {names[b_]:8s} {len(v_):5d} {int(np.median(v_)):7d} {int(np.mean(v_)):7d}  {sum(v_) / tot:6.1%}")


if __name__ == "__main__":
    main()
