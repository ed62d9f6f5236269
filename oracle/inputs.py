"""Seeded synthetic inputs shared by the golden generator, the parity tests and the bench's CPU leg
(TEST INFRASTRUCTURE ONLY).  numpy-only so the same values are produced here, on the GPU box and in
tests/golden/make_golden.py.  All values are bf16-representable fp32, so the bf16 GPU path and the
fp32 oracle see identical numbers.

Weights ~ N(0, 0.02^2) (the shipped init std=1e-4 / zero bias makes every
score ~1e-5 and tie-prone; it is used only as an edge case, ``near_zero_init=True``).
"""
from __future__ import annotations

import numpy as np

from .lis import bf16_round

# (name, D, H, N, seed): the shapes SURVEY.md section 8c lists for the committed goldens.
GOLDEN_CASES = [
    ("tiny", 64, 32, 40, 11),
    ("qwen3b_256", 2048, 1024, 256, 12),
    ("qwen3b_576", 2048, 1024, 576, 13),
    ("qwen7b_2304", 3584, 1792, 2304, 16),
    ("ov8b_5832", 4096, 2048, 5832, 14),
]
BUDGETS = (0.1, 0.2, 0.5)


def make_case(d: int, hd: int, n: int, seed: int, near_zero_init: bool = False, batch: int | None = None):
    """-> dict(h [N,D] or [B,N,D], wq [H,D], bq [H], wk [H,D], bk [H]) fp32, bf16-representable."""
    rng = np.random.default_rng(seed)
    shape = (n, d) if batch is None else (batch, n, d)
    h = bf16_round(rng.standard_normal(shape, dtype=np.float32))
    std = 1e-4 if near_zero_init else 0.02
    wq = bf16_round(std * rng.standard_normal((hd, d), dtype=np.float32))
    wk = bf16_round(std * rng.standard_normal((hd, d), dtype=np.float32))
    if near_zero_init:
        bq = np.zeros(hd, np.float32)
        bk = np.zeros(hd, np.float32)
    else:
        bq = bf16_round(0.02 * rng.standard_normal(hd, dtype=np.float32))
        bk = bf16_round(0.02 * rng.standard_normal(hd, dtype=np.float32))
    return dict(h=h, wq=wq, bq=bq, wk=wk, bk=bk)


def make_vec(n: int, seed: int, scale: float = 1.0) -> np.ndarray:
    return (scale * np.random.default_rng(seed).standard_normal(n, dtype=np.float32)).astype(np.float32)


def make_prompt(n_visual: int, n_pre: int, n_post: int, visual_token_id: int, seed: int,
                vision_start_id: int = 151652, vision_end_id: int = 151653, vocab: int = 1000) -> np.ndarray:
    """A synthetic chat prompt: n_pre text ids, <vision_start>, n_visual visual ids, <vision_end>, n_post text."""
    rng = np.random.default_rng(seed)
    pre = rng.integers(10, vocab, n_pre)
    post = rng.integers(10, vocab, n_post)
    ids = np.concatenate((pre, [vision_start_id], np.full(n_visual, visual_token_id), [vision_end_id], post))
    return ids.astype(np.int64)[None, :]


# ---- attention goldens (tests/golden/attn_eager_*.npz, produced by the reference's own eager Qwen2_5_VLAttention) ----
#        name            lens          hq  hkv  d    causal  seed
ATTN_GOLDEN_CASES = [
    ("small_causal", [37, 53], 4, 2, 16, True, 31),
    ("small_full", [24, 9, 40], 4, 4, 16, False, 32),
    ("d128_causal", [70, 33], 4, 2, 128, True, 33),      # a shape the HIP kernels run as well
    ("d128_full", [64, 64, 20], 2, 1, 128, False, 34),
]


def make_attention_inputs(lens, hq, hkv, d, seed):
    """q [T,Hq,d], k / v [T,Hkv,d], dout [T,Hq,d]: N(0,1) rounded to bf16-representable fp32 (so that a bf16 kernel and the
    fp32 reference see the same numbers)."""
    from . import lis as _l
    rng = np.random.default_rng(seed)
    t = int(sum(lens))
    f = lambda *s: _l.bf16_round(rng.standard_normal(s, dtype=np.float32))  # noqa: E731
    return f(t, hq, d), f(t, hkv, d), f(t, hkv, d), f(t, hq, d)
