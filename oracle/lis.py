"""numpy restatement of the reference's LIS + top-k arithmetic (TEST INFRASTRUCTURE ONLY).

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Abbreviations: FT = qwen-vl-finetune, EV = qwen-evaluation,
OV = llava-ov-15.  The three ``selector_scorer.py`` copies are byte-identical;
FT's is cited.

Parity: pinned against vectors produced by importing the reference modules
(tests/golden/make_golden.py); see tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------------------
# bf16 helpers (the GPU path stores tokens / weights as bf16; the oracle sees the same
# values widened to fp32)
# --------------------------------------------------------------------------------------


def bf16_bits(a: np.ndarray) -> np.ndarray:
    """fp32 -> bf16 bit pattern (uint16), round-to-nearest-even (what torch's .bfloat16() does)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    nan = np.isnan(a)
    bias = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    r = ((u + bias) >> np.uint32(16)).astype(np.uint16)
    if nan.any():
        r = np.where(nan, np.uint16(0x7FC0), r)
    return r


def bf16_to_f32(bits: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(bits, dtype=np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def bf16_round(a: np.ndarray) -> np.ndarray:
    """fp32 values rounded to the nearest bf16-representable fp32."""
    return bf16_to_f32(bf16_bits(a))


# --------------------------------------------------------------------------------------
# A2  TransformerScorer.forward
# --------------------------------------------------------------------------------------


def scorer_reference(x, wq, bq, wk, bk):
    """Reference formulation, fp32.  FT/compression_method/selector_scorer.py:34-55.

    x [B,N,D]; wq,wk [H,D]; bq,bk [H] -> scores [B,N]
    k = Linear_k(x) (:47), q = Linear_q(x) (:48), A = q k^T / hidden_dim**0.5 (:51), mean over last dim (:53).
    """
    x = np.asarray(x, np.float32)
    wq = np.asarray(wq, np.float32)
    wk = np.asarray(wk, np.float32)
    hidden = wq.shape[0]
    k = x @ wk.T + np.asarray(bk, np.float32)
    q = x @ wq.T + np.asarray(bq, np.float32)
    attn = (q @ np.swapaxes(k, -1, -2)) / np.float32(hidden ** 0.5)
    return attn.mean(axis=-1, dtype=np.float32)


def scorer_collapsed(x, wq, bq, wk, bk, dtype=np.float64):
    """Algebraically identical form: mean_j(q_i.k_j) = q_i.kbar, kbar = Wk.mean(x)+bk.

    s_i = (x_i.(Wq^T kbar) + bq.kbar)/sqrt(H).  Used to cross-check the HIP kernels'
    formulation against scorer_reference (hard part 2).
    """
    x = np.asarray(x, dtype)
    wq = np.asarray(wq, dtype)
    wk = np.asarray(wk, dtype)
    bq = np.asarray(bq, dtype)
    bk = np.asarray(bk, dtype)
    hidden = wq.shape[0]
    xbar = x.mean(axis=-2)                      # [B,D]
    kbar = xbar @ wk.T + bk                     # [B,H]
    w = kbar @ wq                               # [B,D]
    c = kbar @ bq                               # [B]
    return (np.einsum("bnd,bd->bn", x, w) + c[:, None]) / dtype(math.sqrt(hidden))


# --------------------------------------------------------------------------------------
# A3  hard top-k select
# --------------------------------------------------------------------------------------


def budget_k_eval(n_tokens: int, budgets: float) -> int:
    """EV/token_compression/selector_model.py:186 -- max(1, int(N * budgets)) on a Python double."""
    return max(1, int(n_tokens * budgets))


def budget_k_train(n_tokens: int, budgets: float) -> int:
    """FT/compression_method/selector_model.py:162 -- int(N * budgets), no clamp."""
    return int(n_tokens * budgets)


def order_keys(scores: np.ndarray) -> np.ndarray:
    """Total order used for selection: uint32 key, larger key = selected first.

    -0.0 == +0.0; every NaN is the greatest value (torch.topk's convention).  Ties between equal
    keys are broken by LOWEST INDEX FIRST (torch.topk leaves tie order unspecified; the golden
    fixtures are tie-free at the k boundary, see SURVEY.md section 7 hard part 1).
    """
    s = np.asarray(scores, np.float32) + np.float32(0.0)
    u = s.view(np.uint32)
    neg = (u >> np.uint32(31)).astype(bool)
    key = np.where(neg, ~u, u | np.uint32(0x80000000))
    key = np.where(np.isnan(s), np.uint32(0xFFFFFFFF), key)
    return key.astype(np.uint32)


def hard_topk_indices(scores: np.ndarray, k: int) -> np.ndarray:
    """idx = topk(scores, k).indices.sort().values   (EV/.../selector_model.py:187-188;
    OV/compression_method/modeling_selector.py:176-177).  scores [N] -> int64 [k] ascending."""
    scores = np.asarray(scores, np.float32)
    n = scores.shape[0]
    key = order_keys(scores).astype(np.int64)
    order = np.lexsort((np.arange(n), -key))     # key descending, then index ascending
    return np.sort(order[:k]).astype(np.int64)


def hard_select(h: np.ndarray, scores: np.ndarray, k: int):
    """out = hidden_states[idx, :]   (EV/.../selector_model.py:189)."""
    idx = hard_topk_indices(scores, k)
    return h[idx, :], idx


def constraint_mask(scores: np.ndarray, k: int) -> np.ndarray:
    """zeros_like(scores).scatter_(topk idx, 1.0)   (FT/.../selector_model.py:168-171)."""
    y = np.zeros(scores.shape[0], np.float32)
    y[hard_topk_indices(scores, k)] = 1.0
    return y


# --------------------------------------------------------------------------------------
# A4/A5  differentiable top-k
# --------------------------------------------------------------------------------------


def _sigmoid32(x):
    x = np.asarray(x, np.float32)
    return (np.float32(1.0) / (np.float32(1.0) + np.exp(-x))).astype(np.float32)


def find_ts(xs: np.ndarray, k: int):
    """_find_ts, fp32.  FT/compression_method/selector_model.py:72-86.  xs [B,N] -> (ts [B,1], ps [B,N])."""
    xs = np.asarray(xs, np.float32)
    b, n = xs.shape
    assert 0 < k < n                                                   # :75
    lo = -xs.max(axis=1, keepdims=True) - np.float32(10)               # :78
    hi = -xs.min(axis=1, keepdims=True) + np.float32(10)               # :79
    kf = np.float32(k)
    for _ in range(64):                                                # :80
        mid = (hi + lo) / np.float32(2)                                # :81
        mask = _sigmoid32(xs + mid).sum(axis=1, dtype=np.float32) < kf  # :82
        lo[mask] = mid[mask]                                           # :83
        hi[~mask] = mid[~mask]                                         # :84
    ts = (lo + hi) / np.float32(2)                                     # :85
    return ts, _sigmoid32(xs + ts)                                     # :86


def find_ts_bf16_reference(xs: np.ndarray, k: int):
    """_find_ts as the reference runs it on a BFLOAT16 score tensor (its released scorers are bf16): the same lines with every
    operation's result rounded to bf16 -- lo / hi / mid, xs + mid, the sigmoid, and .sum() (fp32 inside, bf16 out: spacing 2 at
    256 .. 512), so the comparison with k stalls the bisection on a bf16 neighbour of the root.
    EV/token_compression/selector_model.py:75-89 (= FT/compression_method/selector_model.py:72-86).  xs [B,N] (rounded to bf16
    on entry) -> (ts [B,1], ps [B,N]) float32 holding bf16 values.  Pinned bit for bit on tests/golden/lisbf16_*.npz (`ts_bf16`,
    `ps_bf16` from `scores_bf16`): tests/test_oracle_golden.py."""
    r = bf16_round
    xs = r(np.asarray(xs, np.float32))
    b, n = xs.shape
    assert 0 < k < n                                                   # :77
    lo = r(r(-xs.max(axis=1, keepdims=True)) - np.float32(10))         # :80
    hi = r(r(-xs.min(axis=1, keepdims=True)) + np.float32(10))         # :81
    kf = np.float32(k)
    for _ in range(64):                                                # :82
        mid = r(r(hi + lo) / np.float32(2))                            # :83
        s = r(r(_sigmoid32(r(xs + mid))).sum(axis=1, dtype=np.float32))
        mask = s < kf                                                  # :84
        lo[mask] = mid[mask]                                           # :85
        hi[~mask] = mid[~mask]                                         # :86
    ts = r(r(lo + hi) / np.float32(2))                                 # :87
    return ts, r(_sigmoid32(r(xs + ts)))                               # :88


def soft_topk(xs: np.ndarray, k: int) -> np.ndarray:
    """topk = TopK.apply (forward).  FT/.../selector_model.py:53-58,88."""
    return find_ts(xs, k)[1]


def soft_topk_backward(grad_output, xs, ts):
    """TopK.backward.  FT/.../selector_model.py:60-70.  v = sigmoid'(xs+ts); J = diag(v) - v v^T / sum(v)."""
    p = _sigmoid32(np.asarray(xs, np.float32) + np.asarray(ts, np.float32))
    v = p * (np.float32(1) - p)                                        # :66 (closed form of vmap(grad(sigmoid)))
    s = v.sum(axis=1, keepdims=True, dtype=np.float32)                 # :67
    uv = np.asarray(grad_output, np.float32) * v                       # :69
    t1 = -uv.sum(axis=1, keepdims=True, dtype=np.float32) * v / s      # :70
    return t1 + uv


# --------------------------------------------------------------------------------------
# A6/A7  training mask-apply, constraint mask, BCE constraint loss
# --------------------------------------------------------------------------------------


def bce_mean(p, y):
    """F.binary_cross_entropy(p, y) (mean), with ATen's log clamp at -100.  FT/.../selector_model.py:310."""
    p = np.asarray(p, np.float32)
    y = np.asarray(y, np.float32)
    with np.errstate(divide="ignore"):
        lp = np.maximum(np.log(p), np.float32(-100))
        l1p = np.maximum(np.log(np.float32(1) - p), np.float32(-100))
    return np.float32((-(y * lp + (np.float32(1) - y) * l1p)).mean(dtype=np.float64))


def train_forward(h, wq, bq, wk, bk, budgets: float, out_dtype=np.float32):
    """LIS block of the training vision-tower forward.  FT/.../selector_model.py:158-173.

    h [N,D] -> (h_new [N,D], img_mask ps [N], constraint_img_mask y [N], scores [N], ts)
    """
    h = np.asarray(h, np.float32)
    scores = scorer_reference(h[None], wq, bq, wk, bk)[0]              # :159-160
    n = scores.shape[0]
    k = budget_k_train(n, budgets)                                     # :162
    ts, ps = find_ts(scores[None], k)                                  # :163
    ps = ps[0]
    h_new = ps[:, None] * h                                            # :164-165
    if out_dtype == "bf16":
        h_new = bf16_round(h_new)                                      # :166 .type(hidden_states.dtype)
    y = constraint_mask(scores, k)                                     # :168-171
    return h_new, ps, y, scores, np.float32(ts[0, 0])


def lis_backward_explicit(h, wq, bq, wk, bk, g):
    """Backward of scorer_reference w.r.t. parameters and input, fp64, written out op by op
    (what autograd does through FT/compression_method/selector_scorer.py:47-53).

    g = dL/dscores [N].  Returns dict(dwq, dbq, dwk, dbk, dx).
    """
    h = np.asarray(h, np.float64)
    wq = np.asarray(wq, np.float64)
    wk = np.asarray(wk, np.float64)
    bq = np.asarray(bq, np.float64)
    bk = np.asarray(bk, np.float64)
    g = np.asarray(g, np.float64)
    n = h.shape[0]
    hd = wq.shape[0]
    rs = 1.0 / math.sqrt(hd)
    q = h @ wq.T + bq
    k = h @ wk.T + bk
    d_attn = np.repeat((g / n)[:, None], n, axis=1)        # d mean
    dq = d_attn @ k * rs
    dk = d_attn.T @ q * rs
    return dict(dwq=dq.T @ h, dbq=dq.sum(0), dwk=dk.T @ h, dbk=dk.sum(0), dx=dq @ wq + dk @ wk)


def lis_backward_closed(h, wq, bq, wk, bk, g):
    """Closed form of the same gradients (SURVEY.md section 7 hard part 4); rank-1 weight grads.

    kbar = Wk xbar + bk; dq_i = g_i kbar/sqrt(H); dk_j = (sum_i g_i q_i)/(N sqrt(H)) for every j.
    """
    h = np.asarray(h, np.float64)
    wq = np.asarray(wq, np.float64)
    wk = np.asarray(wk, np.float64)
    bq = np.asarray(bq, np.float64)
    bk = np.asarray(bk, np.float64)
    g = np.asarray(g, np.float64)
    n = h.shape[0]
    rs = 1.0 / math.sqrt(wq.shape[0])
    xbar = h.mean(0)
    kbar = wk @ xbar + bk
    gx = g @ h                                             # sum_i g_i x_i   [D]
    sg = g.sum()
    dk = (wq @ gx + bq * sg) * rs / n                      # [H]  (= sum_i g_i q_i / (N sqrt H))
    xsum = h.sum(0)
    w = wq.T @ kbar
    dx = np.outer(g, w) * rs + (wk.T @ dk)[None, :]
    return dict(dwq=np.outer(kbar * rs, gx), dbq=kbar * rs * sg, dwk=np.outer(dk, xsum), dbk=dk * n, dx=dx)


def train_backward(h, wq, bq, wk, bk, budgets: float, d_hnew, reg_weight: float):
    """Gradient of  L = <d_hnew, h_new> + reg_weight * BCE(ps, y)  through the training LIS block
    (FT/.../selector_model.py:158-173 forward, :308-311 loss, :60-70 TopK.backward), fp64 closed form.

    Returns dict(dwq, dbq, dwk, dbk, dx, dps, dscores).  dx includes the ps*d_hnew term.
    """
    h64 = np.asarray(h, np.float64)
    _, ps, y, scores, ts = train_forward(h, wq, bq, wk, bk, budgets)
    n = h64.shape[0]
    p = ps.astype(np.float64)
    # d BCE / d p with the same clamp semantics as ATen's backward: (p - y) / max((1-p) p, 1e-12) / N
    dbce = (p - y) / np.maximum((1.0 - p) * p, 1e-12) / n
    dps = (np.asarray(d_hnew, np.float64) * h64).sum(1) + reg_weight * dbce
    dscores = soft_topk_backward(dps[None].astype(np.float32), scores[None], np.float32(ts))[0].astype(np.float64)
    out = lis_backward_closed(h, wq, bq, wk, bk, dscores)
    out["dx"] = out["dx"] + p[:, None] * np.asarray(d_hnew, np.float64)
    out["dps"] = dps
    out["dscores"] = dscores
    return out


# --------------------------------------------------------------------------------------
# A8  curriculum annealing of the constraint weight
# --------------------------------------------------------------------------------------


def curriculum_weight(global_step: int, max_steps: int, reg_weight_start: float, reg_weight_end: float) -> float:
    """ScheduledWeightTrainer.compute_loss.  FT/qwenvl/train/train_qwen_selector.py:66-79
    (OV/src/train/train_sft_visionselector.py:38-51)."""
    if max_steps > 0:
        progress = min(global_step / max_steps, 1.0)
        return reg_weight_start + (reg_weight_end - reg_weight_start) * progress
    return reg_weight_start
