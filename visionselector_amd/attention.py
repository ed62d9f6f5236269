"""Var-len attention entry points with the reference's call shapes, backed by vsel_varlen_attn_fwd.

  _flash_attention_forward(...)          qwen-vl-finetune/qwenvl/train/trainer.py:29-120  (attention_mask IS cu_seqlens)
  _update_causal_mask(...)               qwen-vl-finetune/qwenvl/train/trainer.py:123-131
  replace_qwen2_vl_attention_class()     qwen-vl-finetune/qwenvl/train/trainer.py:134-149
  vsel_attention_forward(...)            transformers-5.x AttentionInterface signature (prefill of the *_Selector models;
                                         replaces Qwen2_5_VLFlashAttention2.forward -> flash_attn of
                                         qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py:827-918)
Training: when q / k / v require grad (the reference trains the LIS through the frozen LLM, so every attention layer
back-propagates), the call goes through _VarlenAttnFunction = vsel_varlen_attn_fwd_lse + vsel_varlen_attn_bwd (the role of
FlashAttnVarlenFunc behind trainer.py:101-113).  There is no eager / CPU fallback.
"""
from __future__ import annotations

import weakref
from typing import Optional

import torch

from . import ops

ATTN_NAME = "vsel_varlen"
# Same function under a name that contains "flash": transformers hands var-len metadata (cu_seq_lens_q / max_length_q) to an
# attention interface only when `is_flash_attention_requested(config)` -- e.g. the Qwen2.5-VL vision tower otherwise splits
# the packed windows and calls the interface once per window (144 calls per layer at 1344 x 1344).  Use this name for
# vision towers (head_dim 80 / 64, non-causal windows), where the reference itself calls flash_attn_varlen_func
# (qwen-evaluation/qwen25vl/modeling_qwen2_5_vl.py, Qwen2_5_VLVisionFlashAttention2).
ATTN_NAME_PACKED = "vsel_flash_varlen"


class _VarlenAttnFunction(torch.autograd.Function):
    """q [T,Hq,d], k/v [T,Hkv,d] bf16 contiguous, cu_seqlens int32 [S+1] -> out [T,Hq,d]; saves (q, k, v, out, lse)."""

    @staticmethod
    def forward(ctx, q, k, v, cu_seqlens, max_seqlen, causal, softmax_scale):
        out, lse = ops.varlen_attn_fwd_lse(q, k, v, cu_seqlens, max_seqlen, causal=causal, softmax_scale=softmax_scale)
        ctx.save_for_backward(q, k, v, out, lse, cu_seqlens)
        ctx.meta = (max_seqlen, causal, softmax_scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse, cu_seqlens = ctx.saved_tensors
        max_seqlen, causal, softmax_scale = ctx.meta
        dq, dk, dv = ops.varlen_attn_bwd(dout.contiguous(), q, k, v, out, lse, cu_seqlens, max_seqlen, causal=causal,
                                         softmax_scale=softmax_scale)
        return dq, dk, dv, None, None, None, None


def varlen_attention(q, k, v, cu_seqlens, max_seqlen: int, causal: bool = True, softmax_scale: Optional[float] = None):
    """Differentiable var-len attention: the plain forward kernel under no_grad / for tensors without grad, otherwise the
    LSE-saving forward with the native backward."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        return _VarlenAttnFunction.apply(q, k, v, cu_seqlens, int(max_seqlen), bool(causal), softmax_scale)
    return ops.varlen_attn(q, k, v, cu_seqlens, max_seqlen, causal=causal, softmax_scale=softmax_scale)


def _flash_attention_forward(query_states, key_states, value_states, attention_mask, query_length, is_causal,
                             dropout: float = 0.0, position_ids=None, softmax_scale: Optional[float] = None,
                             sliding_window=None, use_top_left_mask: bool = False, softcap=None, deterministic=None,
                             cu_seq_lens_q=None, cu_seq_lens_k=None, max_length_q=None, max_length_k=None,
                             target_dtype=None, **kwargs):
    """Packed batch: query/key/value [1, T, H, d]; `attention_mask` carries cu_seqlens (data_qwen.py:586-595)."""
    assert query_states.size(0) == key_states.size(0) == value_states.size(0) == 1          # trainer.py:75
    if dropout:
        raise NotImplementedError("attention dropout is not supported")
    if softcap is not None or sliding_window is not None:
        raise NotImplementedError("softcap / sliding_window are not supported")
    q, k, v = (t.squeeze(0).contiguous() for t in (query_states, key_states, value_states))
    cu_seqlens = attention_mask.to(torch.int32).contiguous()                                # :79
    with torch.no_grad():
        max_seqlen = int((cu_seqlens[1:] - cu_seqlens[:-1]).max().item())                  # :81-87 (host sync, as the reference)
    causal = is_causal if not use_top_left_mask else (is_causal and query_length != 1)      # :89-93
    out = varlen_attention(q, k, v, cu_seqlens, max_seqlen, causal=bool(causal), softmax_scale=softmax_scale)
    return out.unsqueeze(0)


def _preload_flash_flavour():
    """Constructing / loading a model WITH attn_implementation="vsel_flash_varlen" makes transformers (>= 5) call
    lazy_import_flash_attention(name) for any name containing "flash", which would look for a hub kernel of that name.
    Mark the name as already loaded, with our flash-attn-compatible functions behind transformers' own flash helpers."""
    try:
        from transformers import modeling_flash_attention_utils as fu
        from . import flash_attn_compat as compat
        if not hasattr(fu, "_loaded_implementation"):
            return
        fu._loaded_implementation = ATTN_NAME_PACKED
        fu._flash_fn, fu._flash_varlen_fn, fu._flash_with_kvcache_fn = compat.flash_attn_func, compat.flash_attn_varlen_func, None
        fu._pad_fn, fu._unpad_fn = fu._pad_input, fu._unpad_input
        fu._process_flash_kwargs_fn = fu._lazy_define_process_function(fu._flash_varlen_fn)
    except Exception:       # older / newer transformers without these internals: set config._attn_implementation after loading
        pass


def _update_causal_mask(self, attention_mask, input_tensor, cache_position, past_key_values, output_attentions):
    return attention_mask                                                                   # trainer.py:123-131


def vsel_attention_forward(module, query, key, value, attention_mask, dropout: float = 0.0, scaling: Optional[float] = None,
                           is_causal: Optional[bool] = None, **kwargs):
    """transformers AttentionInterface function.  query [B, Hq, Lq, d], key/value [B, Hkv, Lk, d].

    Prefill (Lq == Lk): every batch row is one causal sequence -> one var-len call with cu_seqlens = [0, L, 2L, ...]
    (or the packed cu_seq_lens_q kwarg when the batch is flattened).  Decode / chunked prefill (Lq < Lk): the same kernel
    against the cache as one page per batch row (_decode_attention).  Returns (attn_output [B, Lq, Hq, d], None)."""
    if dropout:
        raise NotImplementedError("attention dropout is not supported")
    b, hq, lq, d = query.shape
    lk = key.shape[2]
    if kwargs.get("sliding_window") is not None or kwargs.get("softcap") is not None:
        # the flash-attn call shapes raise for these too (_flash_attention_forward above): never run full attention silently
        raise NotImplementedError("vsel attention does not implement sliding_window / softcap")
    # Batch-1 decode (Lq < Lk): generate() hands a NEW mask object every step, so the padding probe would cost one blocking
    # host sync per generated token (and break stream capture) on the main batch-1 path, for a case the cache path cannot
    # serve anyway -- a left-padded single prompt is not produced by generate() (padding exists to align a BATCH).  Skipped
    # there; documented limit: decode against a padded cache is refused for B > 1 and not detected for B = 1.
    probe = attention_mask is not None and not (b == 1 and lq < lk)
    pad = _padding_info(attention_mask, b, lk) if probe else None
    needs_grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
    if (pad is None and not needs_grad and kwargs.get("cu_seq_lens_q") is None and lq <= lk and query.dtype == torch.bfloat16
            and key.shape == value.shape and all(ops.head_major_ok(t) for t in (query, key, value))):
        # inference, HF's own head-major [B, H, L, d] tensors: no transposing copies (vsel_varlen_attn_fwd_strided); covers the
        # prefill (Lq == Lk) and decode / chunked prefill against the cache (Lq < Lk, bottom-right causal mask)
        causal = True if is_causal is None else bool(is_causal)
        return ops.attn_head_major(query, key, value, causal=causal, softmax_scale=scaling), None
    if lq != lk:
        if pad is not None:
            raise NotImplementedError("vsel attention against the cache does not handle padded batches (keys must be "
                                      "contiguous from position 0): decode one sequence per call or pack the batch")
        return _decode_attention(query, key, value, scaling, is_causal), None
    q = query.transpose(1, 2).reshape(b * lq, hq, d).contiguous()
    k = key.transpose(1, 2).reshape(b * lk, key.shape[1], d).contiguous()
    v = value.transpose(1, 2).reshape(b * lk, value.shape[1], d).contiguous()
    if pad is not None:
        # padded prefill batch (2-D mask, 1 = token): unpad -> one packed var-len call -> pad back with zeros, the
        # flash-attn recipe (bert_padding.unpad_input / pad_input); the padded rows' outputs are zeros
        indices, cu, max_len = pad
        causal = True if is_causal is None else bool(is_causal)
        out_tok = varlen_attention(q[indices].contiguous(), k[indices].contiguous(), v[indices].contiguous(), cu, max_len,
                                   causal=causal, softmax_scale=scaling)
        out = torch.zeros(b * lq, hq, d, dtype=out_tok.dtype, device=out_tok.device).index_copy(0, indices, out_tok)
        return out.view(b, lq, hq, d), None
    cu = kwargs.get("cu_seq_lens_q")
    if cu is not None:
        cu = cu.to(torch.int32).contiguous()
        max_len = int(kwargs.get("max_length_q") or (cu[1:] - cu[:-1]).max().item())
    else:
        cu = torch.arange(0, (b + 1) * lq, lq, dtype=torch.int32, device=query.device)
        max_len = lq
    causal = True if is_causal is None else bool(is_causal)
    out = varlen_attention(q, k, v, cu, max_len, causal=causal, softmax_scale=scaling)
    return out.view(b, lq, hq, d), None


_pad_cache = {}
_decode_meta = {}


def _padding_info(attention_mask, b: int, lk: int):
    """2-D [B, Lk] token mask with at least one padded position -> (token indices [nnz], cu_seqlens int32 [B+1], max length);
    None when there is no padding or the mask is not a 2-D token mask (4-D additive masks are assumed to be plain causal:
    use the flash-flavoured interface name for padded batches).  The result is cached per mask tensor: the 28 layers of one
    forward see the same mask and the check costs one host sync."""
    if attention_mask.dim() != 2 or tuple(attention_mask.shape) != (b, lk):
        return None
    # identity of the tensor OBJECT (all layers of one forward receive the same one) + its version counter; a data_ptr key
    # would go stale when the allocator hands the same block to the next forward's mask
    ref = _pad_cache.get("ref")
    hit = ref is not None and ref() is attention_mask and _pad_cache.get("version") == attention_mask._version
    if not hit:
        mask = attention_mask.to(torch.bool)
        info = None
        if not bool(mask.all()):
            lens = mask.sum(dim=1, dtype=torch.int32)
            cu = torch.nn.functional.pad(torch.cumsum(lens, 0, dtype=torch.int32), (1, 0))
            info = (torch.nonzero(mask.flatten(), as_tuple=False).flatten(), cu.contiguous(), int(lens.max().item()))
        _pad_cache["ref"], _pad_cache["version"], _pad_cache["info"] = weakref.ref(attention_mask), attention_mask._version, info
    return _pad_cache["info"]


def _decode_attention(query, key, value, scaling, is_causal):
    """Lq < Lk (decode / chunked prefill against the cache): every batch row is one sequence whose keys are ONE page of
    Lk rows -> vsel_paged_attn_fwd with page_size = Lk and block_table[b] = [b]; bottom-right aligned causal mask (query i
    sees keys <= i + Lk - Lq), which is what the HF cache layout means.  Inference only."""
    b, hq, lq, d = query.shape
    hkv, lk = key.shape[1], key.shape[2]
    if lq > lk:
        raise ValueError(f"more queries ({lq}) than keys ({lk})")
    if torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad):
        raise RuntimeError("vsel attention against a longer key sequence (decode) has no backward; training uses Lq == Lk")
    q = query.transpose(1, 2).reshape(b * lq, hq, d).contiguous()
    k_pages = key.transpose(1, 2).contiguous()                    # [B pages, Lk, Hkv, d]
    v_pages = value.transpose(1, 2).contiguous()
    dev = query.device
    # the index tensors are the same for every layer of a decode step: build them once per (B, Lq, Lk, device)
    meta_key = (b, lq, lk, dev)
    if _decode_meta.get("key") != meta_key:
        _decode_meta["key"] = meta_key
        _decode_meta["val"] = (torch.arange(0, (b + 1) * lq, lq, dtype=torch.int32, device=dev),
                               torch.full((b,), lk, dtype=torch.int32, device=dev),
                               torch.arange(b, dtype=torch.int32, device=dev).view(b, 1))
    cu_q, seqlens_k, block_table = _decode_meta["val"]
    causal = True if is_causal is None else bool(is_causal)
    out = ops.paged_attn(q, k_pages, v_pages, cu_q, seqlens_k, block_table, lq, causal=causal, softmax_scale=scaling)
    return out.view(b, lq, hq, d)


def replace_qwen2_vl_attention_class():
    """Install the var-len kernel where the reference installs its flash-attn wrapper (trainer.py:134-149).

    transformers >= 4.48 dispatches attention through AttentionInterface: the function is registered under
    `vsel_varlen`; select it with `config._attn_implementation = "vsel_varlen"` (or `attn_implementation=` at load).
    Older module-level hooks are patched too when they exist."""
    import transformers
    from transformers import AttentionInterface
    AttentionInterface.register(ATTN_NAME, vsel_attention_forward)
    AttentionInterface.register(ATTN_NAME_PACKED, vsel_attention_forward)
    try:        # mask flavour: the un-padded (flash) one -- None when the batch has no padding, else the 2-D token mask, which
        #         vsel_attention_forward turns into cu_seqlens; without this a custom interface receives no mask at all
        from transformers.masking_utils import AttentionMaskInterface, flash_attention_mask
        AttentionMaskInterface.register(ATTN_NAME, flash_attention_mask)
        AttentionMaskInterface.register(ATTN_NAME_PACKED, flash_attention_mask)
    except ImportError:      # transformers < 4.53: masks are prepared by the model's _update_causal_mask (patched below)
        pass
    _preload_flash_flavour()
    for mod_name, cls_name in (("qwen2_vl", "Qwen2VLModel"), ("qwen2_5_vl", "Qwen2_5_VLModel")):
        mod = getattr(getattr(transformers.models, mod_name, None), f"modeling_{mod_name}", None)
        if mod is None:
            continue
        if hasattr(mod, "_flash_attention_forward"):
            mod._flash_attention_forward = _flash_attention_forward
        cls = getattr(mod, cls_name, None)
        if cls is not None and hasattr(cls, "_update_causal_mask"):
            cls._update_causal_mask = _update_causal_mask
    return ATTN_NAME
