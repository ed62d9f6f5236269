"""numpy restatement of the sequence splice after selection (TEST INFRASTRUCTURE ONLY).

Follows EV = /root/reference/qwen-evaluation/token_compression/selector_model.py and
OV = /root/reference/llava-ov-15/compression_method/modeling_selector.py.
Batch size is 1 by construction in the reference (EV :270 assert; the image path flattens
``torch.where(...)[1]`` across the batch).
"""
from __future__ import annotations

import numpy as np


def splice_image(input_ids: np.ndarray, image_token_id: int, all_indices: np.ndarray):
    """EV :246-253 (OV modeling_selector.py:262-269).

    input_ids [1,L] int64, all_indices [k] ascending (into the image-token sub-sequence)
    -> (selected_indices [L'], input_ids' [1,L'])
    """
    input_ids = np.asarray(input_ids)
    origin_image_indices = np.where(input_ids == image_token_id)[1]          # :246
    retain_image_indices = origin_image_indices[np.asarray(all_indices)]     # :247
    origin_text_indices = np.where(input_ids != image_token_id)[1]           # :248
    combined = np.concatenate((retain_image_indices, origin_text_indices))   # :249
    selected = np.sort(combined)                                             # :250
    return selected.astype(np.int64), input_ids[:, selected]                 # :253


def splice_embeds(inputs_embeds: np.ndarray, new_input_ids: np.ndarray, selected: np.ndarray,
                  visual_token_id: int, visual_embeds: np.ndarray) -> np.ndarray:
    """EV :254-262: inputs_embeds[:, selected, :] then masked_scatter of the selected visual rows
    into the positions whose (new) id is the visual token, in order."""
    out = np.array(inputs_embeds[:, selected, :], copy=True)
    pos = np.where(new_input_ids[0] == visual_token_id)[0]
    assert pos.shape[0] == visual_embeds.shape[0]
    out[0, pos, :] = visual_embeds
    return out


def splice_video(input_ids: np.ndarray, video_token_id: int, all_indices: np.ndarray,
                 vision_start_id: int = 151652, vision_end_id: int = 151653):
    """EV :264-290 (video branch).

    -> (selected_indices [L'], input_ids' [1,L'], text_image_mask [1,L'])
    """
    input_ids = np.asarray(input_ids)
    assert input_ids.shape[0] == 1                                            # :270
    n_video_tokens = np.asarray(all_indices).shape[0]                         # :267 (= video_embeds.shape[0])
    total_len = input_ids.shape[-1]                                           # :269
    before_idx = int(np.nonzero(input_ids[0] == vision_start_id)[0][0]) + 1   # :271-272
    post_idx = int(np.nonzero(input_ids[0] == vision_end_id)[0][-1])          # :274-275
    vid = np.full((1, n_video_tokens), video_token_id, dtype=input_ids.dtype)  # :277-282
    new_ids = np.concatenate((input_ids[:, :before_idx], vid, input_ids[:, post_idx:]), axis=1)  # :284
    shifted = np.asarray(all_indices) + before_idx                            # :285
    combined = np.concatenate((np.arange(0, before_idx), shifted, np.arange(post_idx, total_len)))  # :286
    selected = np.sort(combined).astype(np.int64)                             # :287
    text_image_mask = new_ids != video_token_id                               # :295
    return selected, new_ids, text_image_mask


def slice_positions(position_ids: np.ndarray, attention_mask: np.ndarray, selected: np.ndarray):
    """EV :318-319: position_ids[:, :, sel] (M-RoPE, [3,1,L], computed from the ORIGINAL ids :311-317)
    and attention_mask[:, sel].  OV :311-314 slices 1-D position_ids / cache_position / attention_mask
    the same way."""
    return position_ids[..., selected], attention_mask[:, selected]


def splice_packed(input_ids: np.ndarray, inputs_embeds: np.ndarray, visual_token_id: int, seq_lens, visual_lens, ks,
                  all_indices: np.ndarray, visual_embeds: np.ndarray, position_ids: np.ndarray | None = None):
    """S prompts packed back to back, each spliced independently with the batch-1 algebra above (EV :246-262, :318-319)
    and concatenated: what running the reference's generation forward once per prompt and packing the results for the
    var-len prefill (FT/qwenvl/train/trainer.py:79-113) gives.  all_indices holds local ranks per prompt.
    -> (selected [T'] packed positions, ids' [T'], embeds' [T', D], position_ids' [R, T'] | None, cu_seqlens' [S+1])."""
    input_ids = np.asarray(input_ids)
    sel_all, ids_all, emb_all, pos_all, cu = [], [], [], [], [0]
    p0 = j0 = 0
    for l_s, n_s, k_s in zip(seq_lens, visual_lens, ks):
        ids = input_ids[None, p0:p0 + l_s]
        assert int((ids == visual_token_id).sum()) == n_s
        idx = np.asarray(all_indices[j0:j0 + k_s], np.int64)
        sel, new_ids = splice_image(ids, visual_token_id, idx)
        emb = splice_embeds(inputs_embeds[None, p0:p0 + l_s], new_ids, sel, visual_token_id, visual_embeds[j0:j0 + k_s])
        sel_all.append(sel + p0)
        ids_all.append(new_ids[0])
        emb_all.append(emb[0])
        if position_ids is not None:
            pos_all.append(position_ids[:, p0:p0 + l_s][:, sel])
        cu.append(cu[-1] + sel.shape[0])
        p0 += l_s
        j0 += k_s
    pos = np.concatenate(pos_all, axis=1) if position_ids is not None else None
    return (np.concatenate(sel_all), np.concatenate(ids_all), np.concatenate(emb_all, axis=0), pos,
            np.asarray(cu, np.int32))
