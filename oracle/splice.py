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
