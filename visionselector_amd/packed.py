"""Packed (padding-free) prefill of a BATCH of image prompts through a Qwen2.5-VL selector model: BASELINE config 5.

The reference's generation forward is batch 1 (`assert ... "selector only support single batch"`,
qwen-evaluation/token_compression/selector_model.py:270).  Here B prompts are served in one pass:
  vision tower once over all images  ->  ragged LIS select, one segment per prompt (the reference's JOINT selection over the
  images of a prompt, k_b = max(1, int(N_b * budgets)), EV :184-189)  ->  vsel_splice_batched (ids / embeds / M-RoPE positions,
  emits cu_seqlens')  ->  the LLM over the packed compressed sequence with var-len attention over cu_seqlens'
  (the packing of qwen-vl-finetune/qwenvl/train/trainer.py:79-113)  ->  last-token logits per prompt.
Per prompt this computes exactly what `Qwen2_5_VLForConditionalGeneration_Selector.forward` computes for that prompt alone
(tests/test_hf_gpu.py::test_packed_prefill_matches_per_prompt_forward).  Prefill only (scoring / first token); data-parallel
serving shards prompts over ranks (ddp.shard_units), no collective.
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import ops
from .attention import ATTN_NAME, ATTN_NAME_PACKED


@torch.no_grad()
def packed_prefill(model, input_ids: Sequence[torch.Tensor], pixel_values: torch.Tensor, image_grid_thw: torch.Tensor,
                   images_per_prompt: Sequence[int], budgets: float | None = None):
    """input_ids: B 1-D int64 tensors (each holding its image placeholder tokens); pixel_values / image_grid_thw: all images of
    the batch back to back in prompt order; images_per_prompt[b] images belong to prompt b (0 allowed).
    -> dict(logits [B, vocab] at each prompt's last token, cu_seqlens int32 [B+1] of the compressed packing,
            kept [B] visual tokens kept per prompt, hidden [T', D_llm])."""
    from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf
    cfg = model.config
    visual = model.model.visual if hasattr(model.model, "visual") else model.visual
    lm = model.model.language_model
    if lm.config._attn_implementation not in (ATTN_NAME, ATTN_NAME_PACKED):
        raise RuntimeError("packed_prefill needs the LLM on the vsel var-len attention "
                           f"(config._attn_implementation = {ATTN_NAME_PACKED!r})")
    if len(images_per_prompt) != len(input_ids) or sum(images_per_prompt) != image_grid_thw.shape[0]:
        raise ValueError("images_per_prompt must have one entry per prompt and sum to the number of image grids")
    dev = pixel_values.device
    budget = float(visual.budgets if budgets is None else budgets)
    merge = visual.spatial_merge_size ** 2
    ids_list = [t.to(dev).reshape(-1) for t in input_ids]
    seq_lens = [int(t.numel()) for t in ids_list]
    tok_per_img = (image_grid_thw.prod(dim=1) // merge).tolist()
    visual_lens, grids, g0 = [], [], 0
    for n_img in images_per_prompt:
        visual_lens.append(int(sum(tok_per_img[g0:g0 + n_img])))
        grids.append(image_grid_thw[g0:g0 + n_img])
        g0 += n_img
    ids = torch.cat(ids_list)
    n_img_tokens = [int((t == cfg.image_token_id).sum().item()) for t in ids_list]
    if n_img_tokens != visual_lens:                                         # the reference's check, FT/.../selector_model.py:210-213
        raise ValueError(f"Image features and image tokens do not match: tokens: {n_img_tokens}, features {visual_lens}")
    ks = [max(1, int(n * budget)) if n > 0 else 0 for n in visual_lens]      # EV :186, per prompt

    # 1. encoder + merger over all images (transformers' own forward; natural token order)
    h = hf.Qwen2_5_VisionTransformerPretrainedModel.forward(visual, pixel_values.type(visual.dtype), image_grid_thw).pooler_output
    # 2. M-RoPE positions of every prompt from its ORIGINAL ids (EV :311-317), packed
    pos = []
    for b, t in enumerate(ids_list):
        mm = torch.zeros_like(t, dtype=torch.int32)
        mm[t == cfg.image_token_id] = 1
        p, _ = model.model.get_rope_index(t[None], mm_token_type_ids=mm[None], image_grid_thw=grids[b] if len(grids[b]) else None,
                                          attention_mask=torch.ones_like(t)[None])
        pos.append(p[:, 0, :])
    pos = torch.cat(pos, dim=1).contiguous()                                 # [3, T]
    emb = model.get_input_embeddings()(ids)
    params = [p.detach().contiguous() for p in visual.importance_scorer.params()]
    if all(n > 0 for n in visual_lens) and h.dtype == emb.dtype and h.shape[1] == emb.shape[1] and getattr(model, "fuse_select_splice", True):
        # 3. ragged LIS (one jointly scored segment per prompt) + packed splice in ONE call: the kept rows go straight from the
        #    merger's output into the packed inputs_embeds' (vsel_lis_select_splice) -- no [sum k, D] tensor in between
        o = ops.lis_select_splice(h.contiguous(), *params, ids, emb.contiguous(), cfg.image_token_id, seq_lens, visual_lens, ks,
                                  position_ids=pos)
        new_ids, new_emb, new_pos, cu = o["input_ids"], o["inputs_embeds"], o["position_ids"], o["cu_seqlens"]
    else:
        # 3. ragged LIS: one segment per prompt that has images; 4. packed splice (prompts without images pass through)
        has = [b for b, n in enumerate(visual_lens) if n > 0]
        out, idx, _ = ops.lis_select_varlen(h.contiguous(), [visual_lens[b] for b in has], [ks[b] for b in has], *params)
        _, new_ids, new_emb, new_pos, cu = ops.splice_batched(ids, emb.contiguous(), cfg.image_token_id, seq_lens, visual_lens, ks,
                                                              idx, out.to(emb.dtype), position_ids=pos)
    # 5. the LLM over the packed compressed sequence (var-len attention over cu)
    max_len = max(l - n + k for l, n, k in zip(seq_lens, visual_lens, ks))
    hidden = lm(inputs_embeds=new_emb[None], position_ids=new_pos[:, None, :], use_cache=False, cu_seq_lens_q=cu, cu_seq_lens_k=cu,
                max_length_q=max_len, max_length_k=max_len).last_hidden_state[0]
    last = cu[1:].to(torch.int64) - 1
    logits = model.lm_head(hidden[last])
    return {"logits": logits, "cu_seqlens": cu, "kept": ks, "hidden": hidden, "input_ids": new_ids}
