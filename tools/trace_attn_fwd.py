#!/usr/bin/env python3
"""Phase timing inside ONE steady-state 64-key tile of the attention forward's main loop (workgroup 0, first item, all waves):
s_memtime stamps (shader clock) at the phase edges.  Needs tools/libvsel_trace.so (python tools/trace_small.py build).
    python tools/trace_attn_fwd.py [N_SEQ L]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from visionselector_amd import _native  # noqa: E402
_native.LIB_PATH = os.environ.get("VSEL_TRACE_LIB", os.path.join(ROOT, "tools", "libvsel_trace.so"))
from visionselector_amd import ops  # noqa: E402

n_seq, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 4096)
hq, hkv = 28, 4
t = n_seq * L
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
cu = torch.arange(0, t + 1, L, dtype=torch.int32, device="cuda")
lib = _native.lib()
lib.vsel_debug_read_fwd_tile_trace.argtypes = [C.c_void_p]
lib.vsel_debug_read_fwd_tile_trace.restype = C.c_int
names = ["issue next tile's direct-to-LDS loads", "S^T = K Q^T (16 MFMA)", "mask + online softmax (VALU)", "O^T += V^T P^T (16 MFMA)",
         "barrier + vmcnt(0)"]
acc = []
for _ in range(5):
    ops.varlen_attn(q, k, v, cu, L)
    torch.cuda.synchronize()
    buf = np.zeros((8, 8), dtype=np.uint64)
    assert lib.vsel_debug_read_fwd_tile_trace(buf.ctypes.data) == 0
    acc.append(buf.astype(np.int64))
a = np.median(np.stack(acc), axis=0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.varlen_attn(q, k, v, cu, L)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"forward {us:.0f} us per launch = {4 * 0.5 * n_seq * L * L * hq * 128 / us * 1e-6:.0f} TFLOP/s (with the stamps compiled in)")
nw = int((a[:, 5] > 0).sum())
print(f"{n_seq} x {L}: cycles per phase of one 64-key tile (median of 5 launches), waves 0..{nw - 1} (w and w + 4 share a SIMD)")
for i, nm in enumerate(names):
    print(f"  {nm:40s}", "  ".join(f"{int(a[w, i + 1] - a[w, i]):6d}" for w in range(nw)))
print(f"  {'tile total':40s}", "  ".join(f"{int(a[w, 5] - a[w, 0]):6d}" for w in range(nw)))
print(f"  {'start relative to wave 0':40s}", "  ".join(f"{int(a[w, 0] - a[0, 0]):6d}" for w in range(nw)))
print("  (16 x v_mfma_f32_32x32x16_bf16 = 512 cycles of the SIMD's matrix pipe; a tile is 32 of them per wave)")
