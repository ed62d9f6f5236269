"""BASELINE config 5 at model level (Qwen2.5-VL-7B geometry, random-init bf16): a dynamic-resolution batch of prompts served
by ONE packed prefill (visionselector_amd.packed.packed_prefill) vs the reference's way -- the batch-1 selector forward prompt
by prompt."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import Qwen2_5_VLConfig  # noqa: E402

from visionselector_amd.attention import ATTN_NAME_PACKED, replace_qwen2_vl_attention_class  # noqa: E402
from visionselector_amd.hf_qwen25vl import Qwen2_5_VLForConditionalGeneration_Selector  # noqa: E402
from visionselector_amd.packed import packed_prefill  # noqa: E402

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
model.model.language_model.config._attn_implementation = ATTN_NAME_PACKED
model.model.visual.config._attn_implementation = ATTN_NAME_PACKED
with torch.no_grad():
    for p in model.visual.importance_scorer.parameters():
        p.copy_(0.02 * torch.randn_like(p))
model.visual.budgets = 0.2

g = torch.Generator().manual_seed(1)
# image sides (in 28-px merged units): 24x24 = 576 ... 64x64 = 4096 visual tokens
sides = [(24, 24), (48, 48), (32, 40), (64, 64), (28, 52), (40, 40), (36, 60), (56, 44)]
prompts, pix, grids = [], [], []
for (a, b) in sides:
    hh, ww = 2 * a, 2 * b
    n_vis = a * b
    t = int(torch.randint(16, 129, (1,), generator=g))
    prompts.append(torch.cat((torch.randint(1000, 5000, (t // 2,), generator=g), torch.tensor([VSTART]), torch.full((n_vis,), IMG),
                              torch.tensor([VEND]), torch.randint(1000, 5000, (t - t // 2,), generator=g))))
    pix.append(torch.randn(hh * ww, 3 * 2 * 14 * 14, generator=g))
    grids.append([1, hh, ww])
pix_all = torch.cat(pix).bfloat16().cuda()
grid_all = torch.tensor(grids).cuda()


def sequential():
    p0 = 0
    outs = []
    for b, ids in enumerate(prompts):
        n_patch = grids[b][1] * grids[b][2]
        model.model.rope_deltas = None
        o = model(input_ids=ids[None].cuda(), attention_mask=torch.ones_like(ids)[None].cuda(), pixel_values=pix_all[p0:p0 + n_patch],
                  image_grid_thw=grid_all[b:b + 1], mm_token_type_ids=(ids == IMG).int()[None].cuda(), logits_to_keep=1)
        outs.append(o.logits[0, -1])
        p0 += n_patch
    return torch.stack(outs)


def packed():
    return packed_prefill(model, prompts, pix_all, grid_all, [1] * len(prompts))["logits"]


def timeit(fn, iters=3):
    with torch.no_grad():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, out


t_seq, l_seq = timeit(sequential)
t_pack, l_pack = timeit(packed)
err = float((l_seq.float() - l_pack.float()).abs().max() / l_seq.float().abs().max())
print(json.dumps({"prompts": len(prompts), "visual_tokens": [a * b for a, b in sides], "budget": 0.2,
                  "sequential_batch1_ms": round(t_seq, 1), "packed_prefill_ms": round(t_pack, 1),
                  "speedup": round(t_seq / t_pack, 2), "max_rel_logit_diff": err}))
