"""LLaVA-OneVision-1.5 model-patch surface of the reference, duck-typed on the model instance.

Reference (llava-ov-15/compression_method/):
  selector_model.py      llavaov15_vision_tower_forward_selector (:56-142), llavaov15_vlmodel_forward_selector (:145-254),
                         llavaov15_generation_forward_selector (:257-372)          -- training, bound with types.MethodType
                         (llava-ov-15/llavaonevision1_5/train/train_sft_visionselector.py:219-225)
  modeling_selector.py   RiceTransformerPretrainedModel_Selector (:68-185), LLaVAOneVision1_5_Model_Selector (:188-336),
                         LLaVAOneVision1_5_ForConditionalGeneration_Selector (:339-352)                 -- inference

The OV model code (Rice ViT, LLaVAOneVision1_5_*) is vendored in the reference tree and is not part of transformers, so
nothing here imports it: the free functions only use what the reference's own versions use on `self` (get_input_embeddings,
get_image_features, language_model, config.image_token_id, lm_head, loss_function, regularization_weight, ...), and the
inference classes are produced by `make_llavaov15_selector_classes(...)` from the three base classes the caller imports
from their OV checkout.  The encoder and the LLM stay the model's own modules; only the LIS block (HIP), the splice (HIP) and
the constraint loss are ours.  tests/test_hf_gpu.py exercises all of it on a stand-in with the same interface.
"""
from __future__ import annotations

import sys
import types
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn.functional as F
from transformers.utils import ModelOutput

from . import ops
from .hf_generic import make_vision_tower_forward_selector
from .selector import TransformerScorer


@dataclass
class VselModelOutputWithPast(ModelOutput):
    """Stand-in for LLaVAOneVision1_5_ModelOutputWithPast when the model's own module does not export one."""
    last_hidden_state: Optional[torch.FloatTensor] = None
    past_key_values: Optional[object] = None
    hidden_states: Optional[Tuple[torch.FloatTensor]] = None
    attentions: Optional[Tuple[torch.FloatTensor]] = None
    rope_deltas: Optional[torch.LongTensor] = None


@dataclass
class VselCausalLMOutputWithPast(ModelOutput):
    """Stand-in for LLaVAOneVision1_5_CausalLMOutputWithPast."""
    loss: Optional[torch.FloatTensor] = None
    logits: Optional[torch.FloatTensor] = None
    past_key_values: Optional[object] = None
    hidden_states: Optional[Tuple[torch.FloatTensor]] = None
    attentions: Optional[Tuple[torch.FloatTensor]] = None
    rope_deltas: Optional[torch.LongTensor] = None


def _output_class(obj, suffix: str, fallback):
    """The model family's own output dataclass (so downstream isinstance checks keep working), else ours."""
    for base in type(obj).__mro__:
        mod = sys.modules.get(base.__module__)
        for name in (dir(mod) if mod is not None else ()):
            cand = getattr(mod, name, None)
            if name.endswith(suffix) and isinstance(cand, type) and issubclass(cand, ModelOutput) \
                    and "rope_deltas" in getattr(cand, "__dataclass_fields__", {}):
                return cand
    return fallback


def _base_tower_forward(self):
    """The tower class's own forward: the selector forward is bound on the INSTANCE (types.MethodType), or defined on a
    subclass produced by make_llavaov15_selector_classes, so the first class forward that is not ours is the encoder."""
    for base in type(self).__mro__:
        fwd = base.__dict__.get("forward")
        if fwd is not None and not getattr(fwd, "_vsel_selector", False):
            return fwd
    raise TypeError(f"{type(self).__name__} has no encoder forward")


def llavaov15_vision_tower_forward_selector(self, hidden_states: torch.Tensor, grid_thw: torch.Tensor,
                                            is_verifying: bool = False):
    """Training tower forward: encoder + merger (the model's own), then soft top-k mask + constraint mask.
    -> (H' [N, D], img_mask [N], constraint_img_mask [N]); `is_verifying` returns the pre-merger states like the reference."""
    base = _base_tower_forward(self)
    if is_verifying:
        return base(self, hidden_states, grid_thw, is_verifying=True)
    return make_vision_tower_forward_selector(base, "train")(self, hidden_states, grid_thw)


llavaov15_vision_tower_forward_selector._vsel_selector = True


def llavaov15_vision_tower_forward_selector_eval(self, hidden_states: torch.Tensor, grid_thw: torch.Tensor,
                                                 is_verifying: bool = False):
    """Inference tower forward (RiceTransformerPretrainedModel_Selector.forward): -> (tokens [k, D], all_indices [k],
    total_token_num); sets last_combined_scores / last_selected_indices."""
    base = _base_tower_forward(self)
    if is_verifying:
        return base(self, hidden_states, grid_thw, is_verifying=True)
    return make_vision_tower_forward_selector(base, "eval")(self, hidden_states, grid_thw)


llavaov15_vision_tower_forward_selector_eval._vsel_selector = True


def _flags(self, output_attentions, output_hidden_states, return_dict):
    cfg = self.config
    return (output_attentions if output_attentions is not None else getattr(cfg, "output_attentions", False),
            output_hidden_states if output_hidden_states is not None else getattr(cfg, "output_hidden_states", False),
            return_dict if return_dict is not None else getattr(cfg, "use_return_dict", True))


def _scatter_features(input_ids, inputs_embeds, token_id: int, feats: torch.Tensor, what: str):
    n_tokens = int((input_ids == token_id).sum().item())
    if n_tokens != feats.shape[0]:                                           # selector_model.py:186-191
        raise ValueError(f"{what} features and {what.lower()} tokens do not match: tokens: {n_tokens}, features {feats.shape[0]}")
    mask = (input_ids == token_id).unsqueeze(-1).expand_as(inputs_embeds).to(inputs_embeds.device)
    return inputs_embeds.masked_scatter(mask, feats.to(inputs_embeds.device, inputs_embeds.dtype))


def _run_language_model(self, inputs_embeds, position_ids, attention_mask, past_key_values, use_cache, output_attentions,
                        output_hidden_states, cache_position):
    outputs = self.language_model(input_ids=None, position_ids=position_ids, attention_mask=attention_mask,
                                  past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache,
                                  output_attentions=output_attentions, output_hidden_states=output_hidden_states,
                                  return_dict=True, cache_position=cache_position)
    cls = _output_class(self, "ModelOutputWithPast", VselModelOutputWithPast)
    return cls(last_hidden_state=outputs.last_hidden_state, past_key_values=outputs.past_key_values,
               hidden_states=getattr(outputs, "hidden_states", None), attentions=getattr(outputs, "attentions", None),
               rope_deltas=getattr(self, "rope_deltas", None))


def _cache_and_positions(past_key_values, use_cache, cache_position, position_ids, inputs_embeds):
    if use_cache and past_key_values is None:
        from transformers import DynamicCache
        past_key_values = DynamicCache()                                     # :224-225
    if cache_position is None:
        seen = past_key_values.get_seq_length() if past_key_values is not None else 0
        cache_position = torch.arange(seen, seen + inputs_embeds.shape[1], device=inputs_embeds.device)   # :227-231
    if position_ids is None:
        position_ids = cache_position.unsqueeze(0)                           # :233-234
    return past_key_values, cache_position, position_ids


def llavaov15_vlmodel_forward_selector(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                                       inputs_embeds=None, use_cache=None, output_attentions=None,
                                       output_hidden_states=None, return_dict=None, pixel_values=None,
                                       pixel_values_videos=None, image_grid_thw=None, video_grid_thw=None, rope_deltas=None,
                                       cache_position=None):
    """Training forward of the VL model (sequence length unchanged): -> (output, img_mask, constraint_img_mask)."""
    output_attentions, output_hidden_states, return_dict = _flags(self, output_attentions, output_hidden_states, return_dict)
    img_mask = constraint_img_mask = None
    if inputs_embeds is None:
        inputs_embeds = self.get_input_embeddings()(input_ids)
        if pixel_values is not None:
            image_embeds, img_mask, constraint_img_mask = self.get_image_features(pixel_values, image_grid_thw)   # :184
            inputs_embeds = _scatter_features(input_ids, inputs_embeds, self.config.image_token_id, image_embeds, "Image")
        if pixel_values_videos is not None:
            video_embeds = self.get_video_features(pixel_values_videos, video_grid_thw)                           # :201
            inputs_embeds = _scatter_features(input_ids, inputs_embeds, self.config.video_token_id, video_embeds, "Video")
        if attention_mask is not None:
            attention_mask = attention_mask.to(inputs_embeds.device)
    past_key_values, cache_position, position_ids = _cache_and_positions(past_key_values, use_cache, cache_position,
                                                                        position_ids, inputs_embeds)
    output = _run_language_model(self, inputs_embeds, position_ids, attention_mask, past_key_values, use_cache,
                                 output_attentions, output_hidden_states, cache_position)
    return (output if return_dict else output.to_tuple()), img_mask, constraint_img_mask                          # :254


def llavaov15_generation_forward_selector(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                                          inputs_embeds=None, labels=None, use_cache=None, output_attentions=None,
                                          output_hidden_states=None, return_dict=None, pixel_values=None,
                                          pixel_values_videos=None, image_grid_thw=None, video_grid_thw=None,
                                          rope_deltas=None, cache_position=None):
    """Training forward of the CausalLM: LM loss + regularization_weight * BCE(img_mask, constraint_img_mask) (:365-367)."""
    output_attentions, output_hidden_states, return_dict = _flags(self, output_attentions, output_hidden_states, return_dict)
    outputs, img_mask, constraint_img_mask = self.model(
        input_ids=input_ids, pixel_values=pixel_values, pixel_values_videos=pixel_values_videos,
        image_grid_thw=image_grid_thw, video_grid_thw=video_grid_thw, position_ids=position_ids,
        attention_mask=attention_mask, past_key_values=past_key_values, inputs_embeds=inputs_embeds, use_cache=use_cache,
        output_attentions=output_attentions, output_hidden_states=output_hidden_states, return_dict=return_dict,
        cache_position=cache_position)
    hidden_states = outputs[0]
    logits = self.lm_head(hidden_states)
    loss = None
    if labels is not None:
        loss = self.loss_function(logits=logits, labels=labels, vocab_size=self.config.vocab_size)
    constraint_loss = None
    if pixel_values is not None and img_mask is not None:
        constraint_loss = F.binary_cross_entropy(img_mask.float(), constraint_img_mask.float())                   # :366
        if loss is not None:
            loss = loss + self.regularization_weight * constraint_loss                                           # :367
    cls = _output_class(self, "CausalLMOutputWithPast", VselCausalLMOutputWithPast)
    out = cls(loss=loss, logits=logits, past_key_values=outputs.past_key_values if return_dict else None,
              hidden_states=getattr(outputs, "hidden_states", None) if return_dict else None,
              attentions=getattr(outputs, "attentions", None) if return_dict else None,
              rope_deltas=getattr(outputs, "rope_deltas", None) if return_dict else None)
    if constraint_loss is not None:
        out.constraint_loss = constraint_loss.detach()
    return out


def install_selector_llavaov15(model, budget: float, in_features: Optional[int] = None, hidden_dim: Optional[int] = None,
                               regularization_weight: float = 0.1):
    """What train_sft_visionselector.py:219-225 does: attach the scorer and bind the three training forwards."""
    vl = model.model
    visual = vl.visual
    p = next(visual.parameters())
    d = in_features or getattr(visual.config, "text_hidden_size", None) or getattr(visual.config, "out_hidden_size")
    visual.budgets = budget
    visual.importance_scorer = TransformerScorer(in_features=d, hidden_dim=hidden_dim or d // 2).to(device=p.device, dtype=p.dtype)
    visual.forward = types.MethodType(llavaov15_vision_tower_forward_selector, visual)
    vl.forward = types.MethodType(llavaov15_vlmodel_forward_selector, vl)
    model.forward = types.MethodType(llavaov15_generation_forward_selector, model)
    model.regularization_weight = regularization_weight
    return model


# --- inference -----------------------------------------------------------------------------------------------------
def llavaov15_vlmodel_forward_selector_eval(self, input_ids=None, attention_mask=None, position_ids=None,
                                            past_key_values=None, inputs_embeds=None, use_cache=None, output_attentions=None,
                                            output_hidden_states=None, return_dict=None, pixel_values=None,
                                            pixel_values_videos=None, image_grid_thw=None, video_grid_thw=None,
                                            rope_deltas=None, cache_position=None):
    """LLaVAOneVision1_5_Model_Selector.forward (modeling_selector.py:202-336): the prefill splices the kept image tokens
    into the sequence -- ids / embeds / 1-D position_ids / attention_mask / cache_position are index-selected (one fused device
    splice instead of where / cat / sort / index / masked_scatter) -- batch 1.  -> (output, visual_token_num)."""
    output_attentions, output_hidden_states, return_dict = _flags(self, output_attentions, output_hidden_states, return_dict)
    visual_token_num = 0
    selected_indices = None
    image_embeds = all_indices = None
    if inputs_embeds is None:
        inputs_embeds = self.get_input_embeddings()(input_ids)
        if pixel_values is not None:
            image_embeds, all_indices, visual_token_num = self.get_image_features(pixel_values, image_grid_thw)  # :245
        if pixel_values_videos is not None:                                   # videos are not compressed by the reference (:278-294)
            video_embeds = self.get_video_features(pixel_values_videos, video_grid_thw)
            inputs_embeds = _scatter_features(input_ids, inputs_embeds, self.config.video_token_id, video_embeds, "Video")
            visual_token_num = int((input_ids == self.config.video_token_id).sum().item())
        if attention_mask is not None:
            attention_mask = attention_mask.to(inputs_embeds.device)
    past_key_values, cache_position, position_ids = _cache_and_positions(past_key_values, use_cache, cache_position,
                                                                        position_ids, inputs_embeds)
    if image_embeds is not None:
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        pos = position_ids.reshape(1, 1, -1).to(torch.int64).contiguous()      # one position row, batch 1
        selected_indices, input_ids, inputs_embeds, pos, attention_mask = ops.splice(
            input_ids.contiguous(), inputs_embeds.contiguous(), self.config.image_token_id, all_indices, image_embeds,
            int(visual_token_num), position_ids=pos, attention_mask=attention_mask.contiguous())
        position_ids = pos.reshape(1, -1)                                      # :311-313
        cache_position = cache_position[selected_indices]                      # :312
    output = _run_language_model(self, inputs_embeds, position_ids, attention_mask, past_key_values, use_cache,
                                 output_attentions, output_hidden_states, cache_position)
    return (output if return_dict else output.to_tuple()), visual_token_num   # :336


llavaov15_vlmodel_forward_selector_eval._vsel_selector = True


def make_llavaov15_selector_classes(rice_tower_cls, vl_model_cls, causal_lm_cls, text_model_cls=None):
    """-> (RiceTransformerPretrainedModel_Selector, LLaVAOneVision1_5_Model_Selector,
    LLaVAOneVision1_5_ForConditionalGeneration_Selector) built on the caller's OV classes."""

    class RiceTransformerPretrainedModel_Selector(rice_tower_cls):
        def __init__(self, config, *args, **kwargs) -> None:
            super().__init__(config, *args, **kwargs)
            d = getattr(config, "text_hidden_size", None) or getattr(config, "out_hidden_size")
            self.importance_scorer = TransformerScorer(in_features=d, hidden_dim=d // 2)          # modeling_selector.py:101
            self.budgets = 1.0
            self.last_combined_scores = None
            self.last_selected_indices = None

        forward = llavaov15_vision_tower_forward_selector_eval

    class LLaVAOneVision1_5_Model_Selector(vl_model_cls):
        base_model_prefix = ""
        _checkpoint_conversion_mapping = {"^model": "language_model"}

        def __init__(self, config):
            super().__init__(config)
            self.visual = RiceTransformerPretrainedModel_Selector._from_config(config.vision_config)
            if text_model_cls is not None:
                self.language_model = text_model_cls._from_config(config.text_config)
            self.rope_deltas = None
            self.post_init()

        forward = llavaov15_vlmodel_forward_selector_eval

    class LLaVAOneVision1_5_ForConditionalGeneration_Selector(causal_lm_cls):
        _checkpoint_conversion_mapping = {"^visual": "model.visual",
                                          r"^model(?!\.(language_model|visual))": "model.language_model"}
        _tied_weights_keys = ["lm_head.weight"]

        def __init__(self, config):
            super().__init__(config)
            self.model = LLaVAOneVision1_5_Model_Selector(config)
            self.lm_head = torch.nn.Linear(config.text_config.hidden_size, config.text_config.vocab_size, bias=False)
            self.post_init()

    return (RiceTransformerPretrainedModel_Selector, LLaVAOneVision1_5_Model_Selector,
            LLaVAOneVision1_5_ForConditionalGeneration_Selector)
