"""Qwen2.5-VL model-patch surface of the reference, rebuilt on the INSTALLED transformers (5.x) classes.

Reference (vendored transformers 4.50 copies, do not construct under 5.x -- SURVEY.md section 7 hard part 6):
  qwen-vl-finetune/compression_method/selector_model.py   qwen25vl_vision_tower_forward_selector (:101-173),
                                                          qwen25vl_generation_forward_selector (:177-326)
  qwen-evaluation/token_compression/selector_model.py     Qwen2_5_VisionTransformerPretrainedModel_Selector (:96-194),
                                                          Qwen2_5_VLForConditionalGeneration_Selector (:196-387)
The encoder (patch embed, window attention blocks, merger, un-reorder) and the LLM are transformers' own modules; only
the LIS block, the splice and the constraint loss are ours.
"""
from __future__ import annotations

import os
import types
from typing import Optional

import torch
import torch.nn.functional as F
from transformers.models.qwen2_5_vl import modeling_qwen2_5_vl as hf

from . import ops
from .hf_generic import make_vision_tower_forward_selector, select_and_splice, select_block_from_tokens
from .selector import TransformerScorer

_BaseTower = hf.Qwen2_5_VisionTransformerPretrainedModel
_BaseCausal = hf.Qwen2_5_VLForConditionalGeneration

# --- training tower forward (bound with types.MethodType, train_qwen_selector.py:191) -----------------------------
qwen25vl_vision_tower_forward_selector = make_vision_tower_forward_selector(_BaseTower.forward, "train")
_tower_forward_eval = make_vision_tower_forward_selector(_BaseTower.forward, "eval")


def _visual_of(model):
    return model.model.visual if hasattr(model, "model") and hasattr(model.model, "visual") else model.visual


def _eval_time() -> bool:
    return os.environ.get("EVAL_TIME", "").lower() == "true"


def qwen25vl_generation_forward_selector(self, input_ids=None, attention_mask=None, position_ids=None,
                                         past_key_values=None, inputs_embeds=None, labels=None, use_cache=None,
                                         pixel_values=None, pixel_values_videos=None, image_grid_thw=None,
                                         video_grid_thw=None, mm_token_type_ids=None, second_per_grid_ts=None, **kwargs):
    """Training forward (sequence length unchanged): visual tokens are multiplied by the soft top-k mask and
    `regularization_weight * BCE(img_mask, constraint_img_mask)` is added to the LM loss (selector_model.py:207-221, :308-311)."""
    visual = _visual_of(self)
    img_mask = constraint_img_mask = None
    if inputs_embeds is None:
        inputs_embeds = self.get_input_embeddings()(input_ids)
        if pixel_values is not None:
            pixel_values = pixel_values.type(visual.dtype)
            image_embeds, img_mask, constraint_img_mask = visual(pixel_values, grid_thw=image_grid_thw)
            n_image_tokens = int((input_ids == self.config.image_token_id).sum().item())
            n_image_features = image_embeds.shape[0]
            if n_image_tokens != n_image_features:                     # selector_model.py:210-213
                raise ValueError(
                    f"Image features and image tokens do not match: tokens: {n_image_tokens}, features {n_image_features}")
            mask = (input_ids == self.config.image_token_id).unsqueeze(-1).expand_as(inputs_embeds)
            inputs_embeds = inputs_embeds.masked_scatter(mask, image_embeds.to(inputs_embeds.device, inputs_embeds.dtype))
            pixel_values = None
    outputs = _BaseCausal.forward(self, input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                                  past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels,
                                  use_cache=use_cache, pixel_values=None, pixel_values_videos=pixel_values_videos,
                                  image_grid_thw=image_grid_thw, video_grid_thw=video_grid_thw,
                                  mm_token_type_ids=mm_token_type_ids, second_per_grid_ts=second_per_grid_ts, **kwargs)
    if labels is not None and img_mask is not None:
        constraint_loss = F.binary_cross_entropy(img_mask.float(), constraint_img_mask.float())   # :310
        outputs.loss = outputs.loss + self.regularization_weight * constraint_loss              # :311
        outputs.constraint_loss = constraint_loss.detach()
    return outputs


def install_selector(model, budget: float, in_features: Optional[int] = None, hidden_dim: Optional[int] = None,
                     regularization_weight: float = 0.1):
    """What train_qwen_selector.py:190-200 does, for a stock transformers model: attach the scorer, bind the two forwards."""
    visual = _visual_of(model)
    d = in_features or visual.config.out_hidden_size
    visual.budgets = budget
    visual.importance_scorer = TransformerScorer(in_features=d, hidden_dim=hidden_dim or d // 2).to(
        device=next(visual.parameters()).device, dtype=next(visual.parameters()).dtype)
    visual.forward = types.MethodType(qwen25vl_vision_tower_forward_selector, visual)
    model.forward = types.MethodType(qwen25vl_generation_forward_selector, model)
    model.regularization_weight = regularization_weight
    return model


# --- inference classes -------------------------------------------------------------------------------------------
class Qwen2_5_VisionTransformerPretrainedModel_Selector(_BaseTower):
    """EV/token_compression/selector_model.py:96-194.  forward -> (tokens [k, D], all_indices [k] int64 ascending,
    total_token_num); sets last_combined_scores [N] and last_selected_indices [k]; reads self.budgets."""

    def __init__(self, config, *inputs, **kwargs) -> None:
        super().__init__(config, *inputs, **kwargs)
        self.importance_scorer = TransformerScorer(in_features=config.out_hidden_size,
                                                   hidden_dim=config.out_hidden_size // 2)      # :124
        self.budgets = 1.0
        self.last_combined_scores = None
        self.last_selected_indices = None

    def forward(self, hidden_states: torch.Tensor, grid_thw: torch.Tensor, **kwargs):
        return _tower_forward_eval(self, hidden_states, grid_thw, **kwargs)


def _tower_call_is_plain(visual) -> bool:
    """True when `visual(...)` would do nothing but run our tower forward, so the fused prefill may call the tower's
    pieces directly.  `nn.Module.__call__` also runs forward / pre-forward hooks (accelerate's device_map hook, profilers)
    and an instance-level `forward` (accelerate's `add_hook_to_module`, `types.MethodType` patches) overrides the class one:
    with any of those present the caller must go through `self.visual(...)`."""
    if type(visual).forward is not Qwen2_5_VisionTransformerPretrainedModel_Selector.forward:
        return False
    if "forward" in visual.__dict__:
        return False
    return not (visual._forward_hooks or visual._forward_pre_hooks
                or torch.nn.modules.module._global_forward_hooks or torch.nn.modules.module._global_forward_pre_hooks)


class Qwen2_5_VLForConditionalGeneration_Selector(_BaseCausal):
    """EV/token_compression/selector_model.py:196-387: prefill splices the kept visual tokens into the sequence
    (ids / embeds / M-RoPE positions / attention mask are index-selected), decode uses the cached rope deltas."""

    def __init__(self, config):
        super().__init__(config)
        self.model.visual = Qwen2_5_VisionTransformerPretrainedModel_Selector._from_config(config.vision_config)
        self.post_init()

    @property
    def visual(self):
        return self.model.visual

    def forward(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None, inputs_embeds=None,
                labels=None, use_cache=None, pixel_values=None, pixel_values_videos=None, image_grid_thw=None,
                video_grid_thw=None, mm_token_type_ids=None, second_per_grid_ts=None, **kwargs):
        timing = _eval_time()
        if timing:
            start = torch.cuda.Event(enable_timing=True)
            start.record()
        visual_token_num = None
        selected_indices = None
        prefill = inputs_embeds is None and (pixel_values is not None or pixel_values_videos is not None)
        # a forward on an empty cache starts a new request: whatever the previous (compressed) prompt left behind --
        # the dropped-column count used to trim generate()'s mask during decode, the video text/visual mask -- is stale
        cache_len = 0 if past_key_values is None else int(past_key_values.get_seq_length())
        if cache_len == 0:
            self._n_dropped = 0
            _set_text_image_mask(self.model.language_model, None)
        if prefill:
            origin_input_ids = input_ids
            inputs_embeds = self.get_input_embeddings()(input_ids)
            # M-RoPE index from the ORIGINAL ids (:311-317); sliced together with ids / embeds / mask by the splice kernel
            full_pos = self.model.compute_3d_position_ids(
                input_ids=origin_input_ids, image_grid_thw=image_grid_thw, video_grid_thw=video_grid_thw,
                inputs_embeds=None, attention_mask=attention_mask, past_key_values=None,
                second_per_grid_ts=second_per_grid_ts, mm_token_type_ids=mm_token_type_ids)
            if full_pos is None:       # no mm_token_type_ids from the processor: derive them from the ids
                full_pos, deltas = self.model.get_rope_index(
                    origin_input_ids, image_grid_thw=image_grid_thw, video_grid_thw=video_grid_thw,
                    attention_mask=attention_mask, second_per_grid_ts=second_per_grid_ts,
                    mm_token_type_ids=_mm_types_from_ids(origin_input_ids, self.config))
                self.model.rope_deltas = deltas
            if pixel_values is not None:
                pix, grid, vis_id = pixel_values.type(self.visual.dtype), image_grid_thw, self.config.image_token_id
            else:
                pix, grid, vis_id = pixel_values_videos.type(self.visual.dtype), video_grid_thw, self.config.video_token_id
            check = getattr(self, "check_token_count", True)     # ValueError on a placeholder / feature count mismatch (FT :210-213)
            am = None if attention_mask is None else attention_mask.contiguous()
            fused, tokens = None, None
            if getattr(self, "fuse_select_splice", True) and _tower_call_is_plain(self.visual):
                # tower -> scores -> hard top-k -> splice with the kept rows written ONCE, from the merger's output straight into
                # inputs_embeds' (vsel_lis_select_splice); bit-identical to visual() + vsel_splice below
                fused, tokens = select_and_splice(self.visual, _BaseTower.forward, pix, grid, input_ids.contiguous(),
                                                  inputs_embeds.contiguous(), vis_id, position_ids=full_pos.contiguous(),
                                                  attention_mask=am, check=check)
            if fused is not None:
                selected_indices, input_ids, inputs_embeds, position_ids, attention_mask, visual_token_num = fused
            else:
                if tokens is not None:             # tower already ran; the LLM has another width / dtype than the tokens
                    vis_embeds, all_indices, visual_token_num = select_block_from_tokens(self.visual, *tokens)
                else:
                    vis_embeds, all_indices, visual_token_num = self.visual(pix, grid_thw=grid)
                # one fused device splice (vsel_splice) instead of where / cat / sort / index / masked_scatter (:246-262, :264-290)
                selected_indices, input_ids, inputs_embeds, position_ids, attention_mask = ops.splice(
                    input_ids.contiguous(), inputs_embeds.contiguous(), vis_id, all_indices, vis_embeds, visual_token_num,
                    position_ids=full_pos.contiguous(), attention_mask=am, check=check)
            if pixel_values_videos is not None and pixel_values is None:
                _set_text_image_mask(self.model.language_model, input_ids != vis_id)        # :295-298
            self._n_dropped = origin_input_ids.shape[1] - input_ids.shape[1]
            # decode positions are (uncompressed length + t) + rope_deltas in the reference (:322-334, cache_position counts
            # the ORIGINAL prompt); transformers 5.x derives them from the cache length L' + t, so fold L - L' into the deltas
            if self.model.rope_deltas is not None:
                self.model.rope_deltas = self.model.rope_deltas + self._n_dropped
            pixel_values = pixel_values_videos = None
            input_ids = None
        elif attention_mask is not None and attention_mask.dim() == 2 and cache_len > 0 and getattr(self, "_n_dropped", 0):
            # decode step: generate() keeps the un-compressed mask (length L + t); the cache holds L' + t positions
            attention_mask = attention_mask[:, self._n_dropped:]
        outputs = _BaseCausal.forward(self, input_ids=input_ids, attention_mask=attention_mask, position_ids=position_ids,
                                      past_key_values=past_key_values, inputs_embeds=inputs_embeds, labels=labels,
                                      use_cache=use_cache, pixel_values=pixel_values, pixel_values_videos=pixel_values_videos,
                                      image_grid_thw=image_grid_thw, video_grid_thw=video_grid_thw,
                                      mm_token_type_ids=None if prefill else mm_token_type_ids,
                                      second_per_grid_ts=second_per_grid_ts, **kwargs)
        if timing and outputs.logits.shape[1] != 1:
            end = torch.cuda.Event(enable_timing=True)
            end.record()
            torch.cuda.synchronize()
            print(f"Input visual token number is: {visual_token_num}")            # :357
            print(f"Generation prefill time is: {start.elapsed_time(end)}")       # :358
        return outputs


def _register_checkpoint_layouts():
    """Released VisionSelector checkpoints carry the transformers-4.50 key layout (`visual.*`, `model.layers.*`).  transformers
    5.x renames such keys for its own classes through a conversion mapping looked up by class name / model_type, but skips
    classes defined outside the library unless their mapping is registered by the user -- so register the stock Qwen2.5-VL
    mapping for the drop-in class."""
    try:
        from transformers import conversion_mapping as cm
        stock = cm.get_checkpoint_conversion_mapping("Qwen2_5_VLForConditionalGeneration") or \
            cm.get_checkpoint_conversion_mapping("qwen2_5_vl")
        if stock is not None:
            cm.register_checkpoint_conversion_mapping("Qwen2_5_VLForConditionalGeneration_Selector", stock, overwrite=True)
    except Exception:        # transformers without the conversion registry (4.5x): the class attribute mapping is inherited
        pass


_register_checkpoint_layouts()


def _set_text_image_mask(language_model, mask):
    """EV/token_compression/selector_model.py:295-298: the video branch publishes the text-vs-visual mask of the compressed
    sequence on the text model AND on every decoder layer's self_attn."""
    language_model.text_image_mask = mask
    for layer in getattr(language_model, "layers", ()):
        attn = getattr(layer, "self_attn", None)
        if attn is not None:
            attn.text_image_mask = mask


def _mm_types_from_ids(input_ids, config):
    t = torch.zeros_like(input_ids, dtype=torch.int32)
    t[input_ids == config.image_token_id] = 1
    t[input_ids == config.video_token_id] = 2
    return t
