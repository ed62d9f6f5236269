#!/usr/bin/env python3
"""Attention kernel throughput vs problem size (Qwen2.5-VL-7B geometry: 28 q heads, 4 kv heads, d = 128, causal)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native
if "--lib" in sys.argv:            # another build of the library (same-box A/B runs: tools/ab_attn.sh)
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from visionselector_amd import ops
rows = []
shapes = [(1, 524), (1, 2368), (8, 524), (32, 524), (4, 2368), (16, 2368), (16, 4096), (4, 8192)]
if "--big" in sys.argv:
    shapes = [(16, 2368), (16, 4096), (4, 8192)]
for nseq, L in shapes:
    g = torch.Generator(device="cuda").manual_seed(7)
    T = nseq * L
    q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
    k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
    cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
    for _ in range(3):
        ops.varlen_attn(q, k, v, cu, L)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = 20
    e0.record()
    for _ in range(it):
        ops.varlen_attn(q, k, v, cu, L)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / it
    fl = 4.0 * L * L * 28 * 128 / 2 * nseq
    rows.append({"n_seq": nseq, "L": L, "us": ms * 1e3, "TFLOPs": fl / (ms * 1e-3) / 1e12})
    print(rows[-1])
