#!/usr/bin/env python3
"""In-kernel timeline of the compressed-prefill attention forward at the metric's own shape (one sequence of 524 tokens,
28 q / 4 kv heads, head_dim 128).  Needs tools/libvsel_trace.so (python tools/trace_small.py build).

    python tools/trace_attn.py [L]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libvsel_trace.so")


def main():
    import numpy as np
    import torch
    from visionselector_amd import _native
    _native.LIB_PATH = LIB
    from visionselector_amd import ops

    L = int(sys.argv[1]) if len(sys.argv) > 1 else 524
    n_seq = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    hq, hkv, d = 28, 4, 128
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(n_seq * L, hq, d, device="cuda", generator=g).bfloat16()
    k = torch.randn(n_seq * L, hkv, d, device="cuda", generator=g).bfloat16()
    v = torch.randn(n_seq * L, hkv, d, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, n_seq * L + 1, L, device="cuda", dtype=torch.int32)
    lib = _native.lib()
    if len(sys.argv) > 2 and sys.argv[2] == "q128":
        lib.vsel_debug_set(_native.KNOBS["attn_split_q64"], 0, None)      # two 4-wave groups, 128 queries per workgroup
    lib.vsel_debug_read_attn_trace.argtypes = [C.c_void_p, C.c_int]
    lib.vsel_debug_read_attn_trace.restype = C.c_int
    buf = np.zeros((8, 1024, 8), dtype=np.uint64)
    f = lambda: ops.varlen_attn(q, k, v, cu, L, causal=True)  # noqa: E731
    for _ in range(20):
        f()
    torch.cuda.synchronize()
    runs = []
    for _ in range(20):
        for _ in range(4):
            f()
        torch.cuda.synchronize()
        assert lib.vsel_debug_read_attn_trace(buf.ctypes.data, 1) == 0
        runs.append(buf.copy())
    import time
    t0 = time.perf_counter()
    for _ in range(2000):
        f()
    torch.cuda.synchronize()
    print(f"{n_seq} x L = {L}: {(time.perf_counter() - t0) / 2000 * 1e6:.2f} us per launch (traced build, back to back)")
    if n_seq > 1:
        # persistent grid: the stamps are those of the LAST item each workgroup processed
        r = runs[-1]
        live = r[0][:, 0] > 0
        st = r[0][live].astype(np.int64)
        t_first = (st[:, 1] - st[:, 0]) / 100.0
        t_loop = (st[:, 2] - st[:, 1]) / 100.0
        t_end = (st[:, 4] - st[:, 2]) / 100.0
        print(f"  last item of each of the {int(live.sum())} workgroups: start -> first tile landed median {np.median(t_first):.2f} us, "
              f"tile loop {np.median(t_loop):.2f}, merge + store {np.median(t_end):.2f}")
        return
    names = ["start", "Q in registers, first tile landed", "tile loop done", "states merged", "end (stores drained)"]
    live = runs[0][0][:, 0] > 0
    nwg = int(live.sum())
    print(f"{nwg} workgroups; us after the first workgroup's start (median over {len(runs)} launches)")
    # workgroups sorted by their number of rounds: report the longest chain (last q-tile) and the shortest separately
    def col(r, kern, slot):
        t0_ = r[0][live][:, 0].min()
        return (r[kern][live][:, slot].astype(np.int64) - int(t0_)) / 100.0
    for s_, nm in enumerate(names):
        x = np.median(np.stack([col(r, 0, s_) for r in runs]), axis=0)
        print(f"  {nm:36s} first {x.min():6.2f}  median {np.median(x):6.2f}  last {x.max():6.2f}")
    print("  rounds (two 64-key tiles per round, one per 4-wave group), workgroup with the longest chain:")
    ends = np.median(np.stack([col(r, 0, 2) for r in runs]), axis=0)
    wg = int(np.argmax(ends))
    for rd in range(8):
        x = np.median(np.stack([col(r, 1, rd) for r in runs]), axis=0)
        if runs[0][1][live][wg, rd] > 0:
            li = np.median(np.stack([col(r, 3, rd) for r in runs]), axis=0)
            bd = np.median(np.stack([col(r, 2, rd) for r in runs]), axis=0)
            print(f"    round {rd}: next tile's loads issued {li[wg]:6.2f}, tile math done {bd[wg]:6.2f}, barrier passed {x[wg]:6.2f}")
    for s_, nm in enumerate(names):
        x = np.median(np.stack([col(r, 0, s_) for r in runs]), axis=0)
        print(f"    {nm:34s} {x[wg]:6.2f}")


if __name__ == "__main__":
    main()
