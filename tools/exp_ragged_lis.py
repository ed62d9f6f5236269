#!/usr/bin/env python3
"""Per-kernel times of the ragged LIS select (config 5: B prompts with N_i ~ U{576..4096} visual tokens, k_i = 0.2 N_i) next to a
uniform batch of the same token count: where does the ragged call lose against the uniform one?  HIP events of the library's profiler.
argv: values of knob lis_seg_sums to try (minimum (segment, slab) pairs of sweep 1's one-wave-per-pair form: more pairs = narrower slabs)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N, ops
d, hd = 3584, 1792
g = torch.Generator(device="cuda").manual_seed(0)
wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]
KN = [int(x) for x in sys.argv[1:]] or [None]
for b in (32, 64, 128):
    rng = np.random.default_rng(b)
    n_vis = [int(x) for x in rng.integers(576, 4097, b)]
    mean = sum(n_vis) // b
    mild = [int(x) for x in rng.integers(2000, 2601, b)]          # max / mean ~ 1.13
    mid = [int(x) for x in rng.integers(1500, 3001, b)]           # max / mean ~ 1.33
    for tag, ns in (("ragged", n_vis), ("uniform", [mean] * b), ("mild", mild), ("mid", mid)):
        ks = [int(n * 0.2) for n in ns]
        h = torch.randn(sum(ns), d, device="cuda", generator=g).bfloat16()
        for kn in KN:
            ctx = N.debug_knob(lis_seg_sums=kn) if kn is not None else N.debug_knob()
            with ctx:
                for _ in range(3):
                    ops.lis_select_varlen(h, ns, ks, wq, bq, wk, bk)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    ops.lis_select_varlen(h, ns, ks, wq, bq, wk, bk)
                e1.record()
                torch.cuda.synchronize()
                N.profile_start()
                for _ in range(10):
                    ops.lis_select_varlen(h, ns, ks, wq, bq, wk, bk)
                prof = N.profile_stop()
            print(json.dumps({"prompts": b, "batch": tag, "tokens": sum(ns), "max_n": max(ns), "lis_seg_sums": kn, "call_us": round(e0.elapsed_time(e1) / 10 * 1e3, 1),
                              "kernels_us": {k.replace("_kernel", ""): round(ms / c * 1e3, 1) for k, (ms, c) in prof.items()}}), flush=True)
        del h
