#!/usr/bin/env python3
"""Throughput sweep over the BASELINE.json configurations (tokens scored+selected per second, one MI355X)."""
import json
import sys
import time

import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops  # noqa: E402


def run(f, iters=20, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def weights(d, hd, g):
    mk = lambda *s: (0.02 * torch.randn(*s, device="cuda", generator=g)).bfloat16()  # noqa: E731
    return mk(hd, d), mk(hd), mk(hd, d), mk(hd)


rows = []
g = torch.Generator(device="cuda").manual_seed(0)
for name, d, hd, n, b, budgets in [("qwen2.5-vl-7b N=2304", 3584, 1792, 2304, 128, (0.1, 0.2, 0.5)),
                                   ("qwen2.5-vl-3b N=576", 2048, 1024, 576, 512, (0.2,)),
                                   ("qwen2.5-vl-3b N=256", 2048, 1024, 256, 1024, (0.2,)),
                                   ("llava-ov-1.5-8b 8x729 joint", 4096, 2048, 5832, 48, (0.2,))]:
    h = torch.randn(b, n, d, device="cuda", generator=g).bfloat16()
    wq, bq, wk, bk = weights(d, hd, g)
    for r in budgets:
        k = max(1, int(n * r))
        t = run(lambda: ops.lis_select(h, wq, bq, wk, bk, k))
        by = b * n * d * 2 + b * k * d * 2 + 2 * hd * d * 2 + b * n * 4 + b * k * 8
        rows.append({"config": name, "B": b, "N": n, "D": d, "k": k, "us_per_step": t * 1e6, "Mtok_per_s": b * n / t / 1e6,
                     "path_GBps": by / t / 1e9})
    del h
# config 5: dynamic-resolution batch, per-segment budgets (ragged)
d, hd = 3584, 1792
wq, bq, wk, bk = weights(d, hd, g)
import random
random.seed(1)
lens = [random.choice([576, 1024, 1600, 2304, 3136, 4096]) for _ in range(96)]
ks = [max(1, int(x * 0.2)) for x in lens]
h = torch.randn(sum(lens), d, device="cuda", generator=g).bfloat16()
t = run(lambda: ops.lis_select_varlen(h, lens, ks, wq, bq, wk, bk))
rows.append({"config": "qwen2.5-vl-7b ragged 96 images N in {576..4096}", "B": 96, "N": sum(lens) / 96, "D": d, "k": sum(ks) / 96,
             "us_per_step": t * 1e6, "Mtok_per_s": sum(lens) / t / 1e6,
             "path_GBps": (sum(lens) * d * 2 + sum(ks) * d * 2 + 2 * hd * d * 2) / t / 1e9})
print(json.dumps(rows, indent=1))
