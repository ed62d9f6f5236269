#!/usr/bin/env python3
"""Why sweep 1 (colsum_partial) reads 310 us on some boxes and 360 on others while sweep 2 holds ~309: per-kernel HIP-event times
of back-to-back vsel_lis_select steps (B = 128) as they are, with an idle gap behind every step's gather (the gather's write-back
drains in the gap instead of under the next sweep 1), and without the gather (scores only).  python tools/exp_sweep1_drain.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from visionselector_amd import _native  # noqa: E402
if "--lib" in sys.argv:
    _native.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
from visionselector_amd import ops  # noqa: E402

b, n, d, hd, k = 128, 2304, 3584, 1792, 460
g = torch.Generator(device="cuda").manual_seed(0)
h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
wq = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
wk = (0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16()
bq = torch.zeros(hd, device="cuda").bfloat16()
bk = bq.clone()


def run(name, step, gap_us=0):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    _native.profile_start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(12):
        step()
        if gap_us:
            torch.cuda._sleep(int(gap_us * 2100))          # ~2.1 GHz shader clock: cycles of an idle kernel on the same stream
    e1.record()
    torch.cuda.synchronize()
    p = _native.profile_stop()
    t = {kk.split("<")[0].replace("vsel::", ""): round(v[0] / v[1] * 1e3, 1) for kk, v in p.items()
         if any(s in kk for s in ("colsum_partial", "colsum_seg", "score_kernel", "gather_rows"))}
    print(f"{name:44s} step {e0.elapsed_time(e1) / 12 * 1e3 - gap_us:7.1f} us (gap excluded)  {t}")


run("back to back (the bench's loop)", lambda: ops.lis_select(h, wq, bq, wk, bk, k))
run("200 us idle behind every step", lambda: ops.lis_select(h, wq, bq, wk, bk, k), gap_us=200)
run("scores only (no select, no gather)", lambda: ops.lis_scores(h, wq, bq, wk, bk))
run("back to back again", lambda: ops.lis_select(h, wq, bq, wk, bk, k))
