#!/usr/bin/env python3
"""Phase timing inside ONE steady-state tile of the dK/dV attention-backward kernel (workgroup 0, all four waves): s_memtime stamps
(shader clock) at the phase edges.  Needs tools/libvsel_trace.so (python tools/trace_small.py build).  python tools/trace_attn_bwd.py [N_SEQ L]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from visionselector_amd import _native  # noqa: E402
_native.LIB_PATH = os.path.join(ROOT, "tools", "libvsel_trace.so")
from visionselector_amd import ops  # noqa: E402

n_seq, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (16, 4096)
hq, hkv = 28, 4
t = n_seq * L
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
do = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
cu = torch.arange(0, t + 1, L, dtype=torch.int32, device="cuda")
out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu, L)
lib = _native.lib()
lib.vsel_debug_read_bwd_trace.argtypes = [C.c_void_p]
lib.vsel_debug_read_bwd_trace.restype = C.c_int
W8 = _native.debug_get("attn_bwd_waves") == 8
if W8:           # eight waves (two per SIMD): each wave owns one 32-query half of the tile
    names = ["S, dP of the wave's half (16 MFMA)", "P, dS (VALU)", "dV, dK (16 MFMA)", "tile tail", "barrier + vmcnt(0)"]
else:
    names = ["S, dP of both sub-blocks (32 MFMA)", "P, dS of sub-block 0 (VALU)", "dV, dK += sub-block 0 (16 MFMA)",
             "P, dS of sub-block 1 (VALU)", "dV, dK += sub-block 1 (16 MFMA)", "tile tail", "barrier + vmcnt(0)"]
NW = 8 if W8 else 4
acc = []
for _ in range(5):
    ops.varlen_attn_bwd(do, q, k, v, out, lse, cu, L)
    torch.cuda.synchronize()
    buf = np.zeros((8, 8), dtype=np.uint64)
    assert lib.vsel_debug_read_bwd_trace(buf.ctypes.data) == 0
    acc.append(buf.astype(np.int64))
a = np.median(np.stack(acc), axis=0)
print(f"{n_seq} x {L}: cycles per phase of one 64-query tile (median of 5 launches), waves 0..{NW - 1}")
for i, nm in enumerate(names):
    print(f"  {nm:36s}", "  ".join(f"{int(a[w, i + 1] - a[w, i]):6d}" for w in range(NW)))
print(f"  {'tile total':36s}", "  ".join(f"{int(a[w, len(names)] - a[w, 0]):6d}" for w in range(NW)))
print("  (16 x v_mfma_f32_32x32x16_bf16 = 512 cycles of the SIMD's matrix pipe)")
