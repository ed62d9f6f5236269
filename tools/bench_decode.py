"""Decode attention against a paged compressed cache (7B heads: 28 q / 4 kv, d 128, Lq = 1): the GQA-packed single-wave form
(one wave per (kv head, sequence), K/V streamed once per group) vs the per-head form of the prefill kernel."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import _native as N  # noqa: E402
from visionselector_amd import ops  # noqa: E402

hq, hkv, page = 28, 4, 64
g = torch.Generator(device="cuda").manual_seed(0)
for b, lk in ((1, 524), (16, 524), (64, 524), (256, 524), (64, 2368), (256, 2368)):
    pages_per = -(-lk // page)
    n_pages = b * pages_per
    q = torch.randn(b, hq, 128, device="cuda", generator=g).bfloat16()
    kc = torch.randn(n_pages, page, hkv, 128, device="cuda", generator=g).bfloat16()
    vc = torch.randn(n_pages, page, hkv, 128, device="cuda", generator=g).bfloat16()
    cu_q = torch.arange(0, b + 1, dtype=torch.int32, device="cuda")
    klens = torch.full((b,), lk, dtype=torch.int32, device="cuda")
    bt = torch.randperm(n_pages, device="cuda", generator=g).to(torch.int32).view(b, pages_per)
    res = {"batch": b, "Lk": lk, "kv_bytes_MB": round(2 * b * lk * hkv * 128 * 2 / 1e6, 1)}
    for mode, tag in ((0, "per_head_us"), (1, "gqa_packed_us"), (2, "default_us")):
        with N.debug_knob("attn_pack", mode):
            for _ in range(3):
                ops.paged_attn(q, kc, vc, cu_q, klens, bt, 1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.paged_attn(q, kc, vc, cu_q, klens, bt, 1)
            e1.record()
            torch.cuda.synchronize()
        res[tag] = round(e0.elapsed_time(e1) / 50 * 1e3, 1)
    res["speedup"] = round(res["per_head_us"] / res["gqa_packed_us"], 2)
    res["packed_kv_GBps"] = round(res["kv_bytes_MB"] / res["gqa_packed_us"] * 1e3, 0)
    print(json.dumps(res))
