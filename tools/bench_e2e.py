"""End-to-end EVAL_TIME-style measurement at Qwen2.5-VL-7B geometry with random-init bf16 weights (no checkpoints offline):
one 1344x1344 image (9216 patches -> 2304 visual tokens) + 64 text tokens, greedy generation of 32 new tokens, for several
retain budgets.  Uses the drop-in classes exactly as the reference's harness does
(Qwen2_5_VLForConditionalGeneration_Selector + model.visual.budgets, EV/predict.py:74-93); the LLM attention runs through
the registered vsel_varlen kernels (prefill and decode) and the vision tower's packed window attention (head_dim 80,
non-causal, cu_seqlens) through the same kernel under the name vsel_flash_varlen (TOWER_ATTN=sdpa: transformers' own path).
Reports, per budget: vision tower ms (budget-independent), LIS + splice ms (native profile), prefill ms (the quantity the
reference prints as "Generation prefill time"), total latency ms, peak memory."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import Qwen2_5_VLConfig  # noqa: E402

from visionselector_amd import _native as N  # noqa: E402
from visionselector_amd.attention import ATTN_NAME, ATTN_NAME_PACKED, replace_qwen2_vl_attention_class  # noqa: E402
from visionselector_amd.hf_qwen25vl import Qwen2_5_VLForConditionalGeneration_Selector  # noqa: E402

IMG, VID, VSTART, VEND = 151655, 151656, 151652, 151653
replace_qwen2_vl_attention_class()
cfg = Qwen2_5_VLConfig(
    text_config=dict(hidden_size=3584, intermediate_size=18944, num_hidden_layers=28, num_attention_heads=28,
                     num_key_value_heads=4, vocab_size=152064, max_position_embeddings=32768,
                     rope_parameters=dict(rope_type="default", mrope_section=[16, 24, 24], rope_theta=1000000.0)),
    vision_config=dict(depth=32, hidden_size=1280, num_heads=16, intermediate_size=3420, out_hidden_size=3584, patch_size=14,
                       spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[7, 15, 23, 31],
                       in_channels=3),
    image_token_id=IMG, video_token_id=VID, vision_start_token_id=VSTART, vision_end_token_id=VEND)
torch.manual_seed(0)
torch.set_default_dtype(torch.bfloat16)
with torch.device("cuda"):
    model = Qwen2_5_VLForConditionalGeneration_Selector(cfg).eval()
torch.set_default_dtype(torch.float32)
model.model.language_model.config._attn_implementation = ATTN_NAME
tower_impl = os.environ.get("TOWER_ATTN", ATTN_NAME_PACKED)      # "sdpa" = transformers' per-window loop
model.model.visual.config._attn_implementation = tower_impl
with torch.no_grad():      # the shipped init (1e-4) makes all scores ~1e-5: use a spread-out scorer so the top-k is well defined
    for p in model.visual.importance_scorer.parameters():
        p.copy_(0.02 * torch.randn_like(p))

grid = (1, 96, 96)
n_patches, n_vis, n_text = 96 * 96, 96 * 96 // 4, 64
g = torch.Generator().manual_seed(1)
pix = torch.randn(n_patches, 3 * 2 * 14 * 14, generator=g).bfloat16().cuda()
pre = torch.randint(1000, 5000, (n_text // 2,), generator=g)
post = torch.randint(1000, 5000, (n_text - n_text // 2 - 2,), generator=g)
ids = torch.cat((pre, torch.tensor([VSTART]), torch.full((n_vis,), IMG), torch.tensor([VEND]), post))[None].cuda()
inputs = dict(input_ids=ids, attention_mask=torch.ones_like(ids), pixel_values=pix,
              image_grid_thw=torch.tensor([list(grid)]).cuda(), mm_token_type_ids=(ids == IMG).int())
new_tokens = 32
out = {"model": "Qwen2.5-VL-7B geometry, random-init bf16", "params_B": sum(p.numel() for p in model.parameters()) / 1e9,
       "tower_attention": tower_impl, "visual_tokens": n_vis, "text_tokens": int(ids.shape[1] - n_vis), "new_tokens": new_tokens, "budgets": {}}

# vision tower alone (budget-independent)
from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf  # noqa: E402
with torch.no_grad():
    for _ in range(2):
        hf.Qwen2_5_VisionTransformerPretrainedModel.forward(model.visual, pix, inputs["image_grid_thw"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        hf.Qwen2_5_VisionTransformerPretrainedModel.forward(model.visual, pix, inputs["image_grid_thw"])
    torch.cuda.synchronize()
    out["vision_tower_ms"] = (time.perf_counter() - t0) / 3 * 1e3

for budget in (1.0, 0.5, 0.2, 0.1):
    model.visual.budgets = budget
    with torch.no_grad():
        for _ in range(2):
            model.model.rope_deltas = None
            model.generate(**inputs, max_new_tokens=4, do_sample=False)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        # prefill = one forward over the prompt (what the reference times inside forward() when logits.shape[1] != 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        pre_ms = []
        for _ in range(3):
            model.model.rope_deltas = None
            e0.record()
            o = model(**inputs, use_cache=True)
            e1.record()
            torch.cuda.synchronize()
            pre_ms.append(e0.elapsed_time(e1))
        N.profile_start()
        model.model.rope_deltas = None
        model(**inputs, use_cache=True)
        torch.cuda.synchronize()
        prof = N.profile_stop()
        lat = []
        for _ in range(2):
            model.model.rope_deltas = None
            e0.record()
            model.generate(**inputs, max_new_tokens=new_tokens, min_new_tokens=new_tokens, do_sample=False)
            e1.record()
            torch.cuda.synchronize()
            lat.append(e0.elapsed_time(e1))
    sel_us = sum(ms for k, (ms, c) in prof.items() if "attn" not in k) * 1e3
    attn_us = sum(ms for k, (ms, c) in prof.items() if "attn" in k) * 1e3
    out["budgets"][str(budget)] = {"kept_visual_tokens": int(model.visual.last_selected_indices.numel()),
                                   "prefill_len": int(o.logits.shape[1]), "prefill_ms": min(pre_ms),
                                   "lis_select_splice_us": sel_us, "llm_attention_us_28_layers": attn_us,
                                   "latency_ms_%d_new_tokens" % new_tokens: min(lat),
                                   "peak_memory_GB": torch.cuda.max_memory_allocated() / 2 ** 30}
b = out["budgets"]
out["prefill_speedup_20pct_vs_100pct"] = b["1.0"]["prefill_ms"] / b["0.2"]["prefill_ms"]
out["llm_prefill_speedup_excluding_tower_20pct"] = (b["1.0"]["prefill_ms"] - out["vision_tower_ms"]) / \
    (b["0.2"]["prefill_ms"] - out["vision_tower_ms"])
print(json.dumps(out, indent=1))
