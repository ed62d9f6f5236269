"""Host-side mirror of the reference's LIS interface, backed by libvsel.so (HIP, gfx950).

Same names, argument meaning and error behaviour as the reference (paths under the reference repo):
  TransformerScorer            qwen-vl-finetune/compression_method/selector_scorer.py:7-55
  TopK / _find_ts / topk       qwen-vl-finetune/compression_method/selector_model.py:53-88
  lis_train_block              the LIS lines of qwen25vl_vision_tower_forward_selector (:158-173)
  lis_select_block             the LIS lines of Qwen2_5_VisionTransformerPretrainedModel_Selector.forward
                               (qwen-evaluation/token_compression/selector_model.py:182-194)
  curriculum_weight            ScheduledWeightTrainer.compute_loss (qwen-vl-finetune/qwenvl/train/train_qwen_selector.py:66-79)
  (the sequence splice, EV :243-320, is ops.splice / ops.splice_batched -- HIP; its CPU restatement lives in oracle/splice.py)

All arithmetic on visual tokens runs in HIP kernels; there is no eager fallback (CPU tensors raise).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn as nn
from torch.autograd import Function

from . import ops


# --------------------------------------------------------------------------------------------------
# differentiable top-k
# --------------------------------------------------------------------------------------------------
class TopK(Function):
    """selector_model.py:53-70.  forward: ps = sigmoid(xs + ts) with sum(ps) = k; backward: J = diag(v) - v v^T / sum v."""

    @staticmethod
    def forward(ctx, xs: torch.Tensor, k: int):
        if xs.dim() != 2:
            raise ValueError("topk expects xs of shape [B, N]")
        b, n = xs.shape
        assert 0 < k < n                                     # selector_model.py:75
        x32 = xs.detach().float().contiguous()
        ps, ts = ops.soft_topk_fwd(x32, int(k))
        ctx.save_for_backward(x32, ts)
        ctx.in_dtype = xs.dtype
        return ps.to(xs.dtype)

    @staticmethod
    def backward(ctx, grad_output):
        x32, ts = ctx.saved_tensors
        g = ops.soft_topk_bwd(grad_output.float().contiguous(), x32, ts)
        return g.to(ctx.in_dtype), None


@torch.no_grad()
def _find_ts(xs: torch.Tensor, k: int):
    """selector_model.py:72-86 -> (ts [B,1], ps [B,N])."""
    b, n = xs.shape
    assert 0 < k < n
    ps, ts = ops.soft_topk_fwd(xs.float().contiguous(), int(k))
    return ts[:, None].to(xs.dtype), ps.to(xs.dtype)


topk = TopK.apply


# --------------------------------------------------------------------------------------------------
# scorer
# --------------------------------------------------------------------------------------------------
class _ScorerFunction(Function):
    """scores = TransformerScorer(x); backward by the closed form (vsel_lis_scores_bwd), one segment per batch item."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk):
        xc = x.detach().contiguous()
        params = tuple(p.detach().contiguous() for p in (wq, bq, wk, bk))
        scores = ops.lis_scores(xc, *params)
        ctx.save_for_backward(xc, *params)
        ctx.need_dx = x.requires_grad
        ctx.x_dtype = x.dtype
        return scores.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        xc, wq, bq, wk, bk = ctx.saved_tensors
        g = g.float().contiguous()
        x3 = xc if xc.dim() == 3 else xc[None]
        g2 = g if g.dim() == 2 else g[None]
        acc = None
        dxs = []
        for b in range(x3.shape[0]):
            dwq, dbq, dwk, dbk, dh = ops.lis_scores_bwd(g2[b].contiguous(), x3[b], wq, bq, wk, bk, need_dh=ctx.need_dx)
            acc = [dwq, dbq, dwk, dbk] if acc is None else [a + n for a, n in zip(acc, (dwq, dbq, dwk, dbk))]
            dxs.append(dh)
        dx = None
        if ctx.need_dx:
            dx = torch.stack(dxs) if xc.dim() == 3 else dxs[0]
        return (dx, acc[0].to(wq.dtype), acc[1].to(bq.dtype), acc[2].to(wk.dtype), acc[3].to(bk.dtype))


class TransformerScorer(nn.Module):
    """Learnable Importance Scorer.  Same constructor, attributes, parameter names/shapes and init as
    selector_scorer.py:7-31, so released checkpoints (`...importance_scorer.{q_proj,k_proj}.{weight,bias}`) load as-is."""

    def __init__(self, in_features: int, hidden_dim: int = 1792, init_scale: float = 0.0001):
        super().__init__()
        self.in_features = in_features
        self.hidden_dim = hidden_dim
        self.k_proj = nn.Linear(in_features, hidden_dim)
        self.q_proj = nn.Linear(in_features, hidden_dim)
        self._init_near_zero(init_scale)

    def _init_near_zero(self, scale: float = 0.0001):
        nn.init.normal_(self.k_proj.weight, std=scale)
        nn.init.zeros_(self.k_proj.bias)
        nn.init.normal_(self.q_proj.weight, std=scale)
        nn.init.zeros_(self.q_proj.bias)

    def params(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        return self.q_proj.weight, self.q_proj.bias, self.k_proj.weight, self.k_proj.bias

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B, N, D] -> scores [B, N]  (selector_scorer.py:34-55), computed by the collapsed HIP path."""
        if x.dim() != 3:
            raise ValueError("TransformerScorer expects x of shape [B, N, D]")
        return _ScorerFunction.apply(x, *self.params())


# --------------------------------------------------------------------------------------------------
# training LIS block
# --------------------------------------------------------------------------------------------------
_FACTOR_SINK = None       # a ddp.LisFactorSync while LisTrainer(exchange="factors") runs a backward (one process per GPU,
                          # no intra-process threading: SURVEY.md section 8b)


class factor_sink:
    """with factor_sink(sync): ...backward...  -- the LIS block's backward leaves its weight gradients as ONE rank-1-factor
    payload row in `sync` (ddp.LisFactorSync.new_row) instead of two dense [Hd, D] tensors; autograd gets no .grad for the
    scorer parameters.  Both are the closed form of SURVEY.md section 7 hard part 4."""

    def __init__(self, sync):
        self.sync = sync

    def __enter__(self):
        global _FACTOR_SINK
        self._prev, _FACTOR_SINK = _FACTOR_SINK, self.sync
        return self.sync

    def __exit__(self, *exc):
        global _FACTOR_SINK
        _FACTOR_SINK = self._prev
        return False


def active_factor_sink():
    return _FACTOR_SINK


class _LisTrainFunction(Function):
    """(h_new, img_mask) = f(h, scorer params); constraint mask returned non-differentiable.
    FT/compression_method/selector_model.py:158-173."""

    @staticmethod
    def forward(ctx, h, wq, bq, wk, bk, k: int):
        hc = h.detach().contiguous()
        params = tuple(p.detach().contiguous() for p in (wq, bq, wk, bk))
        h_new, ps, y, scores, ts, bce = ops.lis_train_fwd(hc, *params, int(k))
        ctx.save_for_backward(hc, *params, ps, y, scores, ts)
        ctx.need_dh = h.requires_grad
        ctx.mark_non_differentiable(y, scores)
        return h_new, ps, y, scores

    @staticmethod
    def backward(ctx, d_hnew, d_ps, _dy, _dscores):
        hc, wq, bq, wk, bk, ps, y, scores, ts = ctx.saved_tensors
        d_ps = None if d_ps is None else d_ps.float().contiguous()
        if _FACTOR_SINK is not None:
            _, dh = ops.lis_train_bwd_factors(d_hnew.contiguous(), hc, wq, bq, wk, bk, ps, y, scores, ts, d_ps_ext=d_ps,
                                              dl_dbce=0.0, need_dh=ctx.need_dh, out=_FACTOR_SINK.new_row(hc.device))
            return dh, None, None, None, None, None
        dwq, dbq, dwk, dbk, dh = ops.lis_train_bwd(d_hnew.contiguous(), hc, wq, bq, wk, bk, ps, y, scores, ts,
                                                   d_ps_ext=d_ps, dl_dbce=0.0, need_dh=ctx.need_dh)
        return dh, dwq.to(wq.dtype), dbq.to(bq.dtype), dwk.to(wk.dtype), dbk.to(bk.dtype), None


def lis_train_block(hidden_states: torch.Tensor, scorer: TransformerScorer, budgets: float):
    """hidden_states [N, D] (all images of the micro-batch jointly) ->
    (hidden_states_new [N, D], img_mask [N], constraint_img_mask [N])   -- selector_model.py:158-173."""
    total_tokens = hidden_states.shape[0]
    k = int(total_tokens * budgets)                                        # :162 (no clamp; _find_ts asserts 0 < k < n)
    assert 0 < k < total_tokens
    h_new, ps, y, _ = _LisTrainFunction.apply(hidden_states, *scorer.params(), k)
    # img_mask / constraint_img_mask stay fp32 (the reference returns the model dtype; a bf16 soft mask would quantise the
    # BCE constraint to 8 bits).  F.binary_cross_entropy(img_mask, constraint_img_mask) works unchanged.
    return h_new, ps, y


# --------------------------------------------------------------------------------------------------
# inference LIS block
# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def lis_select_block(hidden_states: torch.Tensor, scorer: TransformerScorer, budgets: float,
                     with_soft_scores: bool = True, soft_bf16_reference: bool = False):
    """hidden_states [N, D] -> (hidden_states_new [k, D], all_indices int64 [k] ascending, total_token_num,
    last_combined_scores [N] or None)   -- EV/token_compression/selector_model.py:182-194."""
    total_token_num = hidden_states.shape[0]
    dominant_num = max(1, int(total_token_num * budgets))                  # :186
    out, idx, scores = ops.lis_select(hidden_states.contiguous(), *[p.detach().contiguous() for p in scorer.params()],
                                      dominant_num)
    combined = None
    if with_soft_scores and 0 < dominant_num < total_token_num:            # :190 (visualisation only)
        # soft_bf16_reference (tower attribute `soft_topk_bf16_reference`, default off): last_combined_scores in the reference's own bf16
        # arithmetic -- its stalled bisection, bit for bit on the same scores -- instead of the fp32 root (include/vsel.h)
        combined = ops.soft_topk_fwd(scores[None], dominant_num, bf16_reference=soft_bf16_reference)[0][0].to(hidden_states.dtype)
    return out, idx, total_token_num, combined


# --------------------------------------------------------------------------------------------------
# curriculum annealing of the constraint weight
# --------------------------------------------------------------------------------------------------
def curriculum_weight(global_step: int, max_steps: int, reg_weight_start: float = 0.1, reg_weight_end: float = 3.0) -> float:
    """train_qwen_selector.py:66-79 (defaults of the class: 0.1 -> 3.0; the Qwen script passes 0.1 -> 2.0)."""
    if max_steps > 0:
        progress = min(global_step / max_steps, 1.0)
        return reg_weight_start + (reg_weight_end - reg_weight_start) * progress
    return reg_weight_start
