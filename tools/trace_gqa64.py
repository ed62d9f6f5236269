#!/usr/bin/env python3
"""Cycle accounting of attn_fwd_gqa64_kernel from a trace build: shader cycles per wave and ITEM by phase of the generated body, plus
what the C++ around it costs between two bodies (queue draw, decode, pointers, item barrier).
    python tools/trace_gqa64.py build        # here (no GPU): tools/variants/libvsel_gqa64trace.so
    python tools/trace_gqa64.py [n_seq L]    # on the GPU box (default 32 x 524)"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "variants", "libvsel_gqa64trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    inc = "/tmp/gqa64_trace_body.inc"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_attn_fwd64.py")],
                          env=dict(os.environ, F64_OUT=inc, F64_OPTS="heads=1,xitem=1,epi=1,qearly=1,trace=1", F64_PREFIX="VSEL_GQA64"), stdout=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(ROOT, "tools", "build_variant.sh"), "gqa64trace", "attn_fwd_gqa64.hip", "-DVSEL_GQA64_TRACE",
                           f'-DVSEL_GQA64_BODY="{inc}"'])
    sys.exit(0)
import torch
from visionselector_amd import _native
_native.LIB_PATH = LIB
from visionselector_amd import ops
args = [a for a in sys.argv[1:] if a.isdigit()]
nseq, L = (int(args[0]), int(args[1])) if len(args) == 2 else (32, 524)
g = torch.Generator(device="cuda").manual_seed(7)
T = nseq * L
q = torch.randn(T, 28, 128, device="cuda", generator=g).bfloat16()
k = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
v = torch.randn(T, 4, 128, device="cuda", generator=g).bfloat16()
cu = torch.arange(0, T + 1, L, dtype=torch.int32, device="cuda")
h = _native.lib()
h.vsel_debug_read_gqa64_trace.restype = C.c_int
buf = (C.c_uint * 1024)()
with _native.debug_knob(attn_gqa=1, attn_gqa_form=1):
    for _ in range(10):
        ops.varlen_attn(q, k, v, cu, L)
    torch.cuda.synchronize()
    h.vsel_debug_read_gqa64_trace(buf, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.varlen_attn(q, k, v, cu, L)
    e1.record()
    torch.cuda.synchronize()
    h.vsel_debug_read_gqa64_trace(buf, 1)
c = [sum(buf[16 * w + i] for w in range(64)) for i in range(16)]
items = max(c[13], 1)
waves = 4 * items
names = {0: "top_wait_barrier", 1: "phase_x", 2: "phase_y", 3: "exp_tail", 4: "nonsteady_steps", 5: "prologue_rest", 6: "epilogue",
         8: "prologue_issue", 9: "prologue_wait", 10: "s0", 14: "between_bodies(queue, decode, lse, item barrier)"}
rec = {"n_seq": nseq, "L": L, "us": round(e0.elapsed_time(e1) * 1e3, 1), "items": items, "steady_steps_per_wave_item": round(c[7] / waves, 2)}
tot = sum(c[i] for i in names)
for i, nm in names.items():
    rec[nm + "_cyc_per_wave_item"] = round(c[i] / waves, 1)
rec["total_cyc_per_wave_item"] = round(tot / waves, 1)
rec["share"] = {nm: round(c[i] / tot, 3) for i, nm in names.items()}
rec["kernel_cyc_per_workgroup"] = round(c[12] / min(256, items), 1)
print(json.dumps(rec, indent=1))
