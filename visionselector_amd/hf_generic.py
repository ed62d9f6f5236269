"""Model-agnostic wrappers: turn any vision-tower forward that returns merged tokens [N, D] into the reference's
selector tower forwards (training: soft mask + constraint mask; inference: hard top-k + gather).

Used for Qwen2.5-VL (visionselector_amd/hf_qwen25vl.py) and, by a maintainer, for LLaVA-OneVision-1.5's Rice ViT whose
model code is vendored in the reference (llava-ov-15/llavaonevision1_5/, not shipped here): see INTEGRATION.md.
"""
from __future__ import annotations

from typing import Callable

import torch

from .selector import lis_select_block, lis_train_block


def _merged_tokens(out) -> torch.Tensor:
    """Accept either a plain tensor (transformers 4.5x towers) or a ModelOutput with pooler_output (5.x)."""
    if isinstance(out, torch.Tensor):
        return out
    merged = getattr(out, "pooler_output", None)
    if merged is None:
        raise TypeError(f"vision tower returned {type(out).__name__} without merged tokens (pooler_output)")
    return merged


def make_vision_tower_forward_selector(base_forward: Callable, mode: str):
    """base_forward(self, hidden_states, grid_thw, **kw) -> merged tokens.  mode 'train' -> the reference's
    *_vision_tower_forward_selector of compression_method/selector_model.py (returns (H', img_mask, constraint_img_mask));
    mode 'eval' -> the *_Selector.forward of token_compression/selector_model.py / modeling_selector.py
    (returns (tokens[k, D], all_indices[k], total_token_num) and sets last_combined_scores / last_selected_indices)."""
    if mode not in ("train", "eval"):
        raise ValueError("mode must be 'train' or 'eval'")

    def forward_train(self, hidden_states: torch.Tensor, grid_thw: torch.Tensor, **kwargs):
        merged = _merged_tokens(base_forward(self, hidden_states, grid_thw, **kwargs))
        return lis_train_block(merged, self.importance_scorer, self.budgets)

    def forward_eval(self, hidden_states: torch.Tensor, grid_thw: torch.Tensor, **kwargs):
        merged = _merged_tokens(base_forward(self, hidden_states, grid_thw, **kwargs))
        out, idx, total, combined = lis_select_block(merged.detach(), self.importance_scorer, self.budgets)
        self.last_combined_scores = combined
        self.last_selected_indices = idx
        return out, idx, total

    return forward_train if mode == "train" else forward_eval
