"""torch-CPU restatement of the reference formulation of the inference path (TEST INFRASTRUCTURE ONLY).

This is what the reference itself executes on a CPU (same ATen ops in the same order): it is the
``cpu_baseline`` ("port") leg of bench.py and is cross-checked against the numpy oracle and the goldens
in tests/test_oracle_golden.py.  Never imported by the product path.

  scorer  : /root/reference/qwen-vl-finetune/compression_method/selector_scorer.py:44-55
  select  : /root/reference/qwen-evaluation/token_compression/selector_model.py:184-189
  soft k  : /root/reference/qwen-evaluation/token_compression/selector_model.py:75-89 (also run at inference, :190)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


@torch.no_grad()
def scorer_forward(x, wq, bq, wk, bk):
    hidden_dim = wq.shape[0]
    k = F.linear(x, wk, bk)                                            # :47
    q = F.linear(x, wq, bq)                                            # :48
    attn_weights = torch.matmul(q, k.transpose(-2, -1)) / (hidden_dim ** 0.5)   # :51
    return attn_weights.mean(dim=-1)                                   # :53


@torch.no_grad()
def find_ts(xs, k):
    lo = -xs.max(dim=1, keepdims=True).values - 10
    hi = -xs.min(dim=1, keepdims=True).values + 10
    for _ in range(64):
        mid = (hi + lo) / 2
        mask = torch.sigmoid(xs + mid).sum(dim=1) < k
        lo[mask] = mid[mask]
        hi[~mask] = mid[~mask]
    ts = (lo + hi) / 2
    return ts, torch.sigmoid(xs + ts)


@torch.no_grad()
def select_forward(h, wq, bq, wk, bk, budgets: float, with_soft_scores: bool = False):
    """h [N,D] -> (h_new [k,D], idx [k] ascending, scores [N])   EV/.../selector_model.py:182-194"""
    total = h.shape[0]
    scores = scorer_forward(h.unsqueeze(0), wq, bq, wk, bk).squeeze(0)     # :184-185
    k = max(1, int(total * budgets))                                      # :186
    idx = scores.topk(k, dim=0).indices                                   # :187
    idx = idx.sort().values                                               # :188
    h_new = h[idx, :]                                                     # :189
    if with_soft_scores:
        find_ts(scores.unsqueeze(0), k)                                   # :190 (visualisation only)
    return h_new, idx, scores


@torch.no_grad()
def select_forward_collapsed(h, wq, bq, wk, bk, budgets: float):
    """The same selection with the scorer in its algebraically collapsed form (mean_j q_i.k_j = q_i.kbar,
    oracle/lis.py::scorer_collapsed) -- NOT what the reference executes; reported beside the reference formulation in
    bench.py's cpu_baseline so the GPU/CPU ratio is not credited with the algebra."""
    hidden_dim = wq.shape[0]
    total = h.shape[0]
    kbar = F.linear(h.mean(dim=0), wk, bk)
    w = kbar @ wq
    scores = (h @ w + torch.dot(bq, kbar)) / (hidden_dim ** 0.5)
    k = max(1, int(total * budgets))
    idx = scores.topk(k).indices.sort().values
    return h[idx, :], idx, total
