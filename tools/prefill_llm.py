#!/usr/bin/env python3
"""Whole-LLM prefill of Qwen2.5-VL-7B (random-init weights of the real geometry, bf16) at the compressed length
L' = k + 64 vs the uncompressed L = N + 64, attention through the registered `vsel_varlen` kernel, GEMMs through
PyTorch-ROCm (hipBLASLt).  Prints one JSON line."""
import json
import sys
import time

import torch

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def build_llm(layers=28):
    from transformers import Qwen2_5_VLTextConfig
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    from visionselector_amd.attention import ATTN_NAME, replace_qwen2_vl_attention_class
    replace_qwen2_vl_attention_class()
    cfg = Qwen2_5_VLTextConfig(hidden_size=3584, intermediate_size=18944, num_hidden_layers=layers, num_attention_heads=28,
                               num_key_value_heads=4, vocab_size=152064, max_position_embeddings=32768,
                               rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1000000.0))
    cfg._attn_implementation = ATTN_NAME
    with torch.device("cuda"):
        torch.set_default_dtype(torch.bfloat16)
        model = hf.Qwen2_5_VLTextModel(cfg)
        torch.set_default_dtype(torch.float32)
    return model.eval(), cfg


@torch.no_grad()
def time_prefill(model, L, iters=5):
    x = torch.randn(1, L, 3584, device="cuda", dtype=torch.bfloat16) * 0.02
    pos = torch.arange(L, device="cuda")[None, None, :].expand(3, 1, L).contiguous()
    for _ in range(2):
        model(inputs_embeds=x, position_ids=pos, use_cache=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        model(inputs_embeds=x, position_ids=pos, use_cache=False)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


if __name__ == "__main__":
    t0 = time.time()
    model, cfg = build_llm()
    n_params = sum(p.numel() for p in model.parameters())
    res = {"params_B": n_params / 1e9, "build_s": time.time() - t0, "attn_implementation": cfg._attn_implementation}
    for tag, L in (("retain20", 460 + 64), ("retain10", 230 + 64), ("retain50", 1152 + 64), ("full", 2304 + 64)):
        res[tag] = {"L": L, "prefill_ms": time_prefill(model, L)}
    res["speedup_20pct"] = res["full"]["prefill_ms"] / res["retain20"]["prefill_ms"]
    print(json.dumps(res))
