"""numpy restatement of the prefill attention math (TEST INFRASTRUCTURE ONLY).

PINNED against the reference's in-tree EAGER attention module: tests/golden/make_golden.py runs
/root/reference/qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py::Qwen2_5_VLAttention.forward (:749-800: repeat_kv,
QK^T/sqrt(d) + causal mask, fp32 softmax, PV) and torch autograd through it on seeded q / k / v (selection-matrix
projections, identity rotation) and commits outputs and gradients (tests/golden/attn_eager_*.npz);
tests/test_oracle_golden.py::test_attention_oracle_matches_reference_eager_module checks varlen_attention and
varlen_attention_backward against them.
PARITY UNPINNED against the third-party kernel the reference calls in its flash_attention_2 configuration
(flash_attn==2.7.4.post1 via transformers 4.50/4.53 ``_flash_attention_forward``; call sites
/root/reference/qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:900,
/root/reference/llava-ov-15/llavaonevision1_5/modeling_llavaonevision1_5.py:686 and
/root/reference/qwen-vl-finetune/qwenvl/train/trainer.py:101): flash_attn is not in /root/reference and cannot run here.
The var-len contract follows trainer.py:79-113 (cu_seqlens delimit independent causal sequences; q and k share them).
"""
from __future__ import annotations

import math

import numpy as np


def varlen_attention(q: np.ndarray, k: np.ndarray, v: np.ndarray, cu_seqlens: np.ndarray,
                     causal: bool = True, softmax_scale: float | None = None) -> np.ndarray:
    """q [T,Hq,d], k/v [T,Hkv,d], cu_seqlens [S+1] -> out [T,Hq,d], fp64 math.

    GQA: query head h reads kv head h // (Hq // Hkv)  (repeat_kv, modeling_qwen2_5_vl.py:693-702).
    """
    q = np.asarray(q, np.float64)
    k = np.asarray(k, np.float64)
    v = np.asarray(v, np.float64)
    t, hq, d = q.shape
    hkv = k.shape[1]
    rep = hq // hkv
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(d)   # :783
    out = np.zeros_like(q)
    cu = [int(c) for c in cu_seqlens]
    for s in range(len(cu) - 1):
        a, b = cu[s], cu[s + 1]
        if b <= a:
            continue
        n = b - a
        for h in range(hq):
            kk = k[a:b, h // rep, :]
            vv = v[a:b, h // rep, :]
            w = (q[a:b, h, :] @ kk.T) * scale                                     # :783
            if causal:
                w = np.where(np.tril(np.ones((n, n), bool)), w, -np.inf)          # :785-787
            w = w - w.max(axis=1, keepdims=True)
            p = np.exp(w)
            p /= p.sum(axis=1, keepdims=True)                                     # :795 softmax fp32 (here fp64)
            out[a:b, h, :] = p @ vv                                               # :797
    return out


def paged_attention(q: np.ndarray, k_cache: np.ndarray, v_cache: np.ndarray, cu_seqlens_q: np.ndarray,
                    seqlens_k: np.ndarray, block_table: np.ndarray, causal: bool = True,
                    softmax_scale: float | None = None) -> np.ndarray:
    """Same math with the keys of sequence s gathered from pages block_table[s][:] of k_cache / v_cache
    [n_pages, page_size, Hkv, d]; bottom-right aligned causal mask (query i sees keys <= i + klen - qlen, the
    flash-attn >= 2.1 convention the reference's call sites rely on, trainer.py:89-93); rows without a visible key -> 0."""
    q = np.asarray(q, np.float64)
    t, hq, d = q.shape
    page = k_cache.shape[1]
    hkv = k_cache.shape[2]
    rep = hq // hkv
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(d)
    out = np.zeros_like(q)
    for s in range(len(seqlens_k)):
        a, b = int(cu_seqlens_q[s]), int(cu_seqlens_q[s + 1])
        nq, nk = b - a, int(seqlens_k[s])
        if nq <= 0 or nk <= 0:
            continue
        rows = [int(block_table[s][p // page]) * page + p % page for p in range(nk)]
        kk_all = np.asarray(k_cache, np.float64).reshape(-1, hkv, d)[rows]
        vv_all = np.asarray(v_cache, np.float64).reshape(-1, hkv, d)[rows]
        for h in range(hq):
            w = (q[a:b, h, :] @ kk_all[:, h // rep, :].T) * scale
            if causal:
                vis = np.arange(nk)[None, :] <= (np.arange(nq)[:, None] + (nk - nq))
                w = np.where(vis, w, -np.inf)
            mx = w.max(axis=1, keepdims=True)
            ok = np.isfinite(mx[:, 0])
            p_ = np.exp(w - np.where(np.isfinite(mx), mx, 0.0))
            den = p_.sum(axis=1, keepdims=True)
            p_ = np.where(den > 0, p_ / np.where(den > 0, den, 1.0), 0.0)
            out[a:b, h, :] = np.where(ok[:, None], p_ @ vv_all[:, h // rep, :], 0.0)
    return out


def varlen_attention_backward(q: np.ndarray, k: np.ndarray, v: np.ndarray, cu_seqlens: np.ndarray, dout: np.ndarray,
                              causal: bool = True, softmax_scale: float | None = None):
    """Closed-form gradients of varlen_attention w.r.t. q, k, v (fp64): the autograd of the eager formula
    (/root/reference/qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:777-797) that the reference obtains from
    flash_attn_varlen_func's backward when training through the frozen LLM
    (/root/reference/qwen-vl-finetune/qwenvl/train/trainer.py:101-113).  PARITY UNPINNED against flash_attn (absent);
    tests/test_oracle_golden.py pins this function against torch autograd of the same eager formula.
    -> (dq [T,Hq,d], dk [T,Hkv,d], dv [T,Hkv,d]); GQA groups sum into their kv head (repeat_kv's autograd, :693-702)."""
    q = np.asarray(q, np.float64)
    k = np.asarray(k, np.float64)
    v = np.asarray(v, np.float64)
    dout = np.asarray(dout, np.float64)
    t, hq, d = q.shape
    hkv = k.shape[1]
    rep = hq // hkv
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(d)
    dq, dk, dv = np.zeros_like(q), np.zeros_like(k), np.zeros_like(v)
    cu = [int(c) for c in cu_seqlens]
    for s in range(len(cu) - 1):
        a, b = cu[s], cu[s + 1]
        n = b - a
        if n <= 0:
            continue
        for h in range(hq):
            g = h // rep
            w = (q[a:b, h, :] @ k[a:b, g, :].T) * scale
            if causal:
                w = np.where(np.tril(np.ones((n, n), bool)), w, -np.inf)
            p = np.exp(w - w.max(axis=1, keepdims=True))
            p /= p.sum(axis=1, keepdims=True)
            do = dout[a:b, h, :]
            dv[a:b, g, :] += p.T @ do
            dp = do @ v[a:b, g, :].T
            ds = p * (dp - (dp * p).sum(axis=1, keepdims=True))
            dq[a:b, h, :] = ds @ k[a:b, g, :] * scale
            dk[a:b, g, :] += ds.T @ q[a:b, h, :] * scale
    return dq, dk, dv
