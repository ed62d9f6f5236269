#!/usr/bin/env python3
"""In-kernel timeline of the small-batch LIS path (one image, Qwen2.5-VL-7B geometry).

    python tools/trace_small.py build      # here (no GPU): csrc/*.hip with -DVSEL_TRACE -> tools/libvsel_trace.so
    python tools/trace_small.py [B]        # on the GPU box: run, read the stamps of the last call, print the timeline

-DVSEL_TRACE makes thread 0 of every workgroup leave s_memrealtime stamps (100 MHz, device-wide clock) at the phase edges of
the five kernels (csrc/lis_kernels.h VSEL_STAMP).  The shipped libvsel.so is built without it.
"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "tools", "libvsel_trace.so")

KERNELS = ["sweep 1 (colsum_partial)", "kbar projection (proj_nt_small)", "w projection (proj_nn_small)",
           "sweep 2 (score_small)", "select + gather (select_gather_small)", "select + gather: radix passes"]
SLOTS = [["start", "rows summed", "end (stores drained)"],
         ["start", "xbar slice in LDS", "MFMA done", "end (stores drained)"],
         ["start", "kbar slice in LDS", "MFMA done", "end (stores drained)"],
         ["start", "w in LDS", "c reduced", "end (stores drained)"],
         ["start", "keys in registers", "threshold found", "rows known", "end (stores drained)", "ballots done",
          "wave totals exchanged", "positions written"],
         ["pass bits 7:0 done", "pass bits 15:8 done", "pass bits 23:16 done", "pass bits 31:24 done"]]


def build():
    csrc = os.path.join(ROOT, "visionselector_amd", "csrc")
    out = os.path.join(ROOT, "visionselector_amd", "build", "trace")
    os.makedirs(out, exist_ok=True)
    objs, procs = [], []
    for src in sorted(glob.glob(os.path.join(csrc, "*.hip"))):
        obj = os.path.join(out, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DVSEL_TRACE",
                                       "-c", src, "-o", obj]))
    assert all(p.wait() == 0 for p in procs)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    print(LIB)


def main():
    import numpy as np
    import torch
    from visionselector_amd import _native
    _native.LIB_PATH = LIB
    from visionselector_amd import ops

    b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n, d, hd, k = 2304, 3584, 1792, 460
    g = torch.Generator(device="cuda").manual_seed(0)
    h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
    wq = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
    wk = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
    bq = (0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16()
    bk = (0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16()
    lib = _native.lib()
    lib.vsel_debug_read_trace.argtypes = [C.c_void_p, C.c_int]
    lib.vsel_debug_read_trace.restype = C.c_int
    nk, nb, ns = 8, 1024, 8
    buf = np.zeros((nk, nb, ns), dtype=np.uint64)
    runs = []
    for it in range(60):
        ops.lis_select(h, wq, bq, wk, bk, k)
        if it >= 40:                      # steady state: calls queued back to back
            for _ in range(3):
                ops.lis_select(h, wq, bq, wk, bk, k)
            torch.cuda.synchronize()
            assert lib.vsel_debug_read_trace(buf.ctypes.data, 1) == 0
            runs.append(buf.copy())
    # per kernel: origin = earliest start stamp of sweep 1 in that call; medians over calls
    print(f"B = {b}: timeline of one call in us (100 MHz stamps; median over {len(runs)} calls)")
    rows = []
    for r in runs:
        t0 = r[0][..., 0][r[0][..., 0] > 0].min()
        call = []
        for kern in range(len(KERNELS)):
            live = r[kern][:, 0] > 0
            st = r[kern][live]
            nslot = len(SLOTS[kern])
            first = [(st[:, s].min() - t0) / 100.0 for s in range(nslot)]
            last = [(st[:, s].max() - t0) / 100.0 for s in range(nslot)]
            med = [(np.median(st[:, s]) - t0) / 100.0 for s in range(nslot)]
            call.append((int(live.sum()), first, med, last))
        rows.append(call)
    for kern in range(len(KERNELS)):
        wgs = rows[0][kern][0]
        print(f"\n{KERNELS[kern]}: {wgs} workgroups")
        for s, name in enumerate(SLOTS[kern]):
            f = np.median([c[kern][1][s] for c in rows])
            m = np.median([c[kern][2][s] for c in rows])
            la = np.median([c[kern][3][s] for c in rows])
            print(f"  {name:24s} first workgroup {f:7.2f}   median {m:7.2f}   last {la:7.2f}")
    end = np.median([c[4][3][4] for c in rows])
    print(f"\nfirst start of sweep 1 -> last end of select + gather: {end:.2f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        main()
