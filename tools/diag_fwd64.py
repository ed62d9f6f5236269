#!/usr/bin/env python3
"""where do the 64-rows-per-wave forward and the reference forms differ? (mismatch counts per 32-row block and 32-column d-tile)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native, ops
L = int(sys.argv[1]) if len(sys.argv) > 1 else 256
causal = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(L, 1, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(L, 1, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(L, 1, 128, device="cuda", generator=g).bfloat16()
cu = torch.tensor([0, L], dtype=torch.int32, device="cuda")
outs = {}
for r64 in (0, 1):
    with _native.debug_knob(attn_rows64=r64, attn_split=0):
        outs[r64] = ops.varlen_attn(q, k, v, cu, L, causal=causal).float().cpu()
d = (outs[0] != outs[1])[:, 0, :]
print("rows x dtile mismatch counts (32-row blocks):")
for rb in range((L + 31) // 32):
    blk = d[32 * rb:32 * rb + 32]
    print(rb, [int(blk[:, 32 * dt:32 * dt + 32].sum()) for dt in range(4)], "maxdiff", float((outs[0] - outs[1])[32 * rb:32 * rb + 32].abs().max()))
# fp64 reference to see which one is closer
qq, kk, vv = q[:, 0].double().cpu(), k[:, 0].double().cpu(), v[:, 0].double().cpu()
s = qq @ kk.T / (128 ** 0.5)
if causal:
    s = s.masked_fill(torch.ones(L, L).triu(1).bool(), float("-inf"))
ref = torch.softmax(s, -1) @ vv
for r64 in (0, 1):
    print("rows64" if r64 else "rows32", "max err vs fp64", float((outs[r64][:, 0].double() - ref).abs().max()), "mean", float((outs[r64][:, 0].double() - ref).abs().mean()))
