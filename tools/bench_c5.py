"""BASELINE config 5 at kernel level (Qwen2.5-VL-7B geometry): a dynamic-resolution batch of B prompts with N_i ~ U{576..4096}
visual tokens and T_i ~ U{16..128} text tokens: ragged LIS select (per-prompt k_i = int(0.2 N_i)) -> packed splice (emits
cu_seqlens') -> var-len causal attention over the compressed packed batch (28 layers), next to the attention over the
uncompressed packing.  One GPU; the 8-GPU form shards prompts over ranks with no data-path collective."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visionselector_amd import ops  # noqa: E402

IMG = 151655
d, hd, d_llm, hq, hkv = 3584, 1792, 3584, 28, 4
g = torch.Generator(device="cuda").manual_seed(0)
wq, wk = [(0.02 * torch.randn(hd, d, device="cuda", generator=g)).bfloat16() for _ in range(2)]
bq, bk = [(0.02 * torch.randn(hd, device="cuda", generator=g)).bfloat16() for _ in range(2)]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for b in (8, 32, 64):
    rng = np.random.default_rng(b)
    n_vis = [int(x) for x in rng.integers(576, 4097, b)]
    n_txt = [int(x) for x in rng.integers(16, 129, b)]
    ks = [int(n * 0.2) for n in n_vis]
    seq = [n + t for n, t in zip(n_vis, n_txt)]
    seq_c = [k + t for k, t in zip(ks, n_txt)]
    h = torch.randn(sum(n_vis), d, device="cuda", generator=g).bfloat16()
    ids = torch.cat([torch.cat((torch.randint(10, 1000, (t // 2,)), torch.full((n,), IMG), torch.randint(10, 1000, (t - t // 2,))))
                     for n, t in zip(n_vis, n_txt)]).cuda()
    emb = torch.randn(sum(seq), d_llm, device="cuda", generator=g).bfloat16()
    pos = torch.arange(sum(seq), device="cuda")[None].expand(3, -1).contiguous()
    out, idx, _ = ops.lis_select_varlen(h, n_vis, ks, wq, bq, wk, bk)
    sel, new_ids, new_emb, new_pos, cu_c = ops.splice_batched(ids, emb, IMG, seq, n_vis, ks, idx, out, position_ids=pos, check=True)
    assert cu_c.tolist() == np.concatenate(([0], np.cumsum(seq_c))).tolist()
    t_lis = timeit(lambda: ops.lis_select_varlen(h, n_vis, ks, wq, bq, wk, bk))
    t_spl = timeit(lambda: ops.splice_batched(ids, emb, IMG, seq, n_vis, ks, idx, out, position_ids=pos))
    # what packed.packed_prefill calls: scores + top-k + splice with the kept rows written once into inputs_embeds' (no [sum k, D] tensor)
    fused = ops.lis_select_splice(h, wq, bq, wk, bk, ids, emb, IMG, seq, n_vis, ks, position_ids=pos)
    assert torch.equal(fused["inputs_embeds"], new_emb) and torch.equal(fused["input_ids"], new_ids)
    t_fused = timeit(lambda: ops.lis_select_splice(h, wq, bq, wk, bk, ids, emb, IMG, seq, n_vis, ks, position_ids=pos))

    def attn_time(lens):
        t = sum(lens)
        q = torch.randn(t, hq, 128, device="cuda", generator=g).bfloat16()
        k = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
        v = torch.randn(t, hkv, 128, device="cuda", generator=g).bfloat16()
        cu = torch.tensor(np.concatenate(([0], np.cumsum(lens))), dtype=torch.int32, device="cuda")
        us = timeit(lambda: ops.varlen_attn(q, k, v, cu, max(lens)))
        flops = sum(4.0 * L * L * hq * 128 / 2 for L in lens)
        return us, flops / us / 1e6

    a_c, tf_c = attn_time(seq_c)
    a_f, tf_f = attn_time(seq)
    print(json.dumps({"prompts": b, "visual_tokens": sum(n_vis), "kept": sum(ks), "packed_len": sum(seq), "packed_len_compressed": sum(seq_c),
                      "lis_select_ragged_us": round(t_lis, 1), "visual_tokens_per_s_M": round(sum(n_vis) / t_lis, 1),
                      "splice_batched_us": round(t_spl, 1),
                      "lis_select_splice_fused_us": round(t_fused, 1),
                      "attn_compressed_us_per_layer": round(a_c, 1), "attn_compressed_TFLOPs": round(tf_c, 1),
                      "attn_full_us_per_layer": round(a_f, 1), "attn_full_TFLOPs": round(tf_f, 1),
                      "attn_speedup": round(a_f / a_c, 2),
                      "select+splice+28_layers_attn_ms": round((t_fused + 28 * a_c) / 1e3, 3),
                      "28_layers_attn_uncompressed_ms": round(28 * a_f / 1e3, 3)}))
