"""flash-attn-compatible entry points backed by libvsel (for code that calls the flash_attn package directly).

The reference imports, from flash_attn==2.7.4.post1 (requirements.txt:5):
    flash_attn.flash_attn_varlen_func / flash_attn_func        qwen-evaluation/token_compression/selector_model.py:25,
                                                               qwen-vl-finetune/qwenvl/train/trainer.py:7 (…flash_attn_interface)
    flash_attn.bert_padding.{index_first_axis, pad_input, unpad_input}     qwen-evaluation/token_compression/selector_model.py:26
    flash_attn.layers.rotary.apply_rotary_emb                               qwen-vl-finetune/compression_method/selector_model.py:31
flash_attn has no ROCm build in this image; `visionselector_amd/dropin/flash_attn/` exposes the functions below under the
same module paths so the reference's vendored modeling files import and run unchanged.  Signatures follow the published
flash-attn 2.7 Python API; unsupported options (dropout, sliding window, softcap, alibi, attention probabilities) raise.
The attention math runs in vsel_varlen_attn_* (HIP); the padding / rotary helpers are index and elementwise torch ops.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import ops
from .attention import varlen_attention


def _reject(dropout_p, window_size, softcap, alibi_slopes, return_attn_probs):
    if dropout_p:
        raise NotImplementedError("attention dropout is not supported")
    if tuple(window_size) != (-1, -1):
        raise NotImplementedError("sliding-window attention is not supported")
    if softcap:
        raise NotImplementedError("softcap is not supported")
    if alibi_slopes is not None:
        raise NotImplementedError("alibi is not supported")
    if return_attn_probs:
        raise NotImplementedError("return_attn_probs is not supported")


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                           softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0, alibi_slopes=None,
                           deterministic=False, return_attn_probs=False, block_table=None):
    """q [total_q, H, d], k / v [total_k, Hk, d] (bf16) -> out [total_q, H, d].  Same packing for q and k (the reference's
    call sites): differentiable (vsel_varlen_attn_fwd_lse / _bwd).  Different packings: forward only, bottom-right causal."""
    _reject(dropout_p, window_size, softcap, alibi_slopes, return_attn_probs)
    if block_table is not None:
        raise NotImplementedError("paged KV through flash_attn_varlen_func: call visionselector_amd.ops.paged_attn")
    q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
    cu_q = cu_seqlens_q.to(torch.int32).contiguous()
    needs_grad = torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad)
    same = cu_seqlens_q is cu_seqlens_k or cu_seqlens_q.data_ptr() == cu_seqlens_k.data_ptr()
    if not same and needs_grad:
        same = q.shape[0] == k.shape[0] and bool(torch.equal(cu_seqlens_q, cu_seqlens_k))
        if not same:
            raise RuntimeError("vsel attention with different query / key packings has no backward")
    if same:
        return varlen_attention(q, k, v, cu_q, int(max_seqlen_q), causal=bool(causal), softmax_scale=softmax_scale)
    cu_k = cu_seqlens_k.to(torch.int32).contiguous()
    return ops.varlen_attn_kv(q, k, v, cu_q, cu_k, int(max_seqlen_q), causal=bool(causal), softmax_scale=softmax_scale)


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, window_size=(-1, -1), softcap=0.0,
                    alibi_slopes=None, deterministic=False, return_attn_probs=False):
    """q [B, Sq, H, d], k / v [B, Sk, Hk, d] -> out [B, Sq, H, d]; every batch row is one sequence."""
    _reject(dropout_p, window_size, softcap, alibi_slopes, return_attn_probs)
    b, sq, h, d = q.shape
    sk = k.shape[1]
    dev = q.device
    cu_q = torch.arange(0, (b + 1) * sq, sq, dtype=torch.int32, device=dev)
    cu_k = cu_q if sk == sq else torch.arange(0, (b + 1) * sk, sk, dtype=torch.int32, device=dev)
    out = flash_attn_varlen_func(q.reshape(b * sq, h, d), k.reshape(b * sk, k.shape[2], d), v.reshape(b * sk, v.shape[2], d),
                                 cu_q, cu_k, sq, sk, softmax_scale=softmax_scale, causal=causal)
    return out.view(b, sq, h, d)


# ---- flash_attn.bert_padding ----------------------------------------------------------------------------------------
def index_first_axis(x: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    return x[indices]


def unpad_input(hidden_states: torch.Tensor, attention_mask: torch.Tensor, unused_mask: Optional[torch.Tensor] = None):
    """[B, S, ...] + mask [B, S] (1 = token) -> (tokens [nnz, ...], indices [nnz], cu_seqlens int32 [B+1],
    max_seqlen_in_batch (int), seqlens_in_batch int32 [B])  -- the 5-tuple of flash-attn 2.7."""
    all_masks = attention_mask if unused_mask is None else attention_mask + unused_mask
    seqlens_in_batch = all_masks.sum(dim=-1, dtype=torch.int32)
    used_seqlens_in_batch = attention_mask.sum(dim=-1, dtype=torch.int32)
    indices = torch.nonzero(all_masks.flatten(), as_tuple=False).flatten()
    max_seqlen_in_batch = int(seqlens_in_batch.max().item())
    cu_seqlens = F.pad(torch.cumsum(seqlens_in_batch, dim=0, dtype=torch.int32), (1, 0))
    flat = hidden_states.reshape(hidden_states.shape[0] * hidden_states.shape[1], *hidden_states.shape[2:])
    return flat[indices], indices, cu_seqlens, max_seqlen_in_batch, used_seqlens_in_batch


def pad_input(hidden_states: torch.Tensor, indices: torch.Tensor, batch: int, seqlen: int) -> torch.Tensor:
    """tokens [nnz, ...] -> [batch, seqlen, ...] with zeros at the padded positions."""
    out = torch.zeros(batch * seqlen, *hidden_states.shape[1:], dtype=hidden_states.dtype, device=hidden_states.device)
    out[indices] = hidden_states
    return out.view(batch, seqlen, *hidden_states.shape[1:])


# ---- flash_attn.layers.rotary -------------------------------------------------------------------------------------------
def apply_rotary_emb(x, cos, sin, interleaved=False, inplace=False, seqlen_offsets=0, cu_seqlens=None, max_seqlen=None):
    """x [B, S, H, d] (or [T, H, d] with cu_seqlens), cos / sin [S, rotary_dim / 2]: rotate the first rotary_dim features,
    (x1, x2) -> (x1 cos - x2 sin, x1 sin + x2 cos) with x1 / x2 the two halves (interleaved=False) or the even / odd
    features (interleaved=True)."""
    if cu_seqlens is not None or (not isinstance(seqlen_offsets, int)) or seqlen_offsets:
        raise NotImplementedError("apply_rotary_emb: cu_seqlens / seqlen_offsets are not supported")
    ro = cos.shape[-1] * 2
    s = x.shape[-3]
    c = cos[:s].unsqueeze(-2).to(torch.float32)
    sn = sin[:s].unsqueeze(-2).to(torch.float32)
    xr = x[..., :ro].to(torch.float32)
    if interleaved:
        x1, x2 = xr[..., 0::2], xr[..., 1::2]
        rot = torch.stack((x1 * c - x2 * sn, x1 * sn + x2 * c), dim=-1).flatten(-2)
    else:
        x1, x2 = xr[..., : ro // 2], xr[..., ro // 2:]
        rot = torch.cat((x1 * c - x2 * sn, x1 * sn + x2 * c), dim=-1)
    out = torch.cat((rot.to(x.dtype), x[..., ro:]), dim=-1)
    if inplace:
        x.copy_(out)
        return x
    return out
