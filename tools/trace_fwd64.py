#!/usr/bin/env python3
"""Cycle accounting of the 64-rows-per-wave attention forward from a trace build (tools/ab_fwd64.py build tr:trace=1):
shader cycles per wave, summed by the kernel over all waves and items, divided here by the number of steady steps.

    python tools/trace_fwd64.py [n_seq L] [--variant name]     (default 4 x 4096: the 32-bit sums stay below 2^32)"""
import ctypes as C, glob, json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visionselector_amd import _native
pat = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else "tr*"
lib = sorted(glob.glob(os.path.join(ROOT, "visionselector_amd", "build", "variants", f"libvsel_{pat}.so")))[0]
_native.LIB_PATH = lib
from visionselector_amd import ops
args = [a for a in sys.argv[1:] if a.isdigit()]
nseq, L = (int(args[0]), int(args[1])) if len(args) == 2 else (4, 4096)
g = torch.Generator(device="cuda").manual_seed(7)
T = nseq * L
q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
for _ in range(40):
    ops.varlen_attn(q, k, v, cu, L)
torch.cuda.synchronize()
h = _native.lib()
h.vsel_debug_read_fwd64_trace.restype = C.c_int
buf = (C.c_uint * 1024)()
h.vsel_debug_read_fwd64_trace(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ops.varlen_attn(q, k, v, cu, L)
e1.record()
torch.cuda.synchronize()
h.vsel_debug_read_fwd64_trace(buf, 1)
c = [sum(buf[16 * w + i] for w in range(64)) for i in range(13)]
names = ["top_wait_barrier", "phase_x", "phase_y", "exp_tail", "nonsteady_steps", "prologue_rest", "epilogue", "steady_steps",
         "pro_issue", "pro_wait", "pro_s0", "x"]
n_steady = max(c[7], 1)
items = (L // 256) * 28 * nseq
waves = items * 4
rec = {"lib": os.path.basename(lib), "n_seq": nseq, "L": L, "us": e0.elapsed_time(e1) * 1e3, "steady_steps_per_wave": c[7] / waves}
for i in range(4):
    rec[names[i] + "_cyc_per_step"] = round(c[i] / n_steady, 1)
for i in (4, 5, 6, 8, 9, 10):
    rec[names[i] + "_cyc_per_wave_item"] = round(c[i] / waves, 1)
tot = sum(c[:7]) + sum(c[8:11])
rec["total_cyc_per_wave_item"] = round(tot / waves, 1)
rec["kernel_cyc_per_workgroup"] = round(c[12] / 256, 1)
rec["asm_cyc_per_workgroup_wave"] = round(tot / 4 / 256, 1)
rec["clock_GHz_if_wg_spans_kernel"] = round(c[12] / 256 / (rec["us"] * 1e3), 3)
rec["share"] = {names[i]: round(c[i] / tot, 3) for i in (0, 1, 2, 3, 4, 5, 6, 8, 9, 10)}
print(json.dumps(rec))
