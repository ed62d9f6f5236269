"""Torch-tensor front end of the C-ABI in include/vsel.h.

Every function takes ROCm tensors, passes raw pointers + the current HIP stream to libvsel.so and
returns torch tensors allocated by the caching allocator.  There is no CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import functools
import itertools
from typing import Optional, Sequence, Tuple

import torch

from . import _native as N

_DT = {torch.bfloat16: N.VSEL_BF16, torch.float32: N.VSEL_F32}


def _code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"visionselector_amd supports bfloat16 / float32 tensors, got {t.dtype}") from None


def _dev(*ts: torch.Tensor, strided_ok: bool = False) -> torch.device:
    d = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("visionselector_amd ops run on the GPU only (HIP kernels in libvsel.so); "
                               f"got a tensor on {t.device}.  There is no CPU fallback.")
        if not strided_ok and not t.is_contiguous():
            raise RuntimeError("visionselector_amd ops need contiguous tensors")
        if d is None:
            d = t.device
        elif t.device != d:
            raise RuntimeError(f"tensors on different devices: {d} vs {t.device}")
    return d


def _stream() -> int:
    """HIP stream of the CURRENT device; every public op runs under _device_guard, which makes the tensors' device current."""
    return torch.cuda.current_stream().cuda_stream


def _device_guard(fn):
    """Run `fn` with the device of its first GPU tensor argument (positional or keyword) current (as every ATen op does): libvsel launches on
    torch's current stream of the current device, so tensors on cuda:1 in a process whose current device is cuda:0
    (HF device_map, multi-GPU processes) must switch first -- otherwise the kernels would be enqueued on the wrong GPU."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for a in itertools.chain(args, kwargs.values()):          # positional and keyword tensors alike
            if isinstance(a, torch.Tensor) and a.is_cuda:
                if a.device.index != torch.cuda.current_device():
                    with torch.cuda.device(a.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapper


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _scorer(wq, bq, wk, bk) -> N.Scorer:
    hd, d = wq.shape
    if wk.shape != (hd, d) or bq.shape != (hd,) or bk.shape != (hd,):
        raise ValueError(f"scorer parameter shapes disagree: wq {tuple(wq.shape)} wk {tuple(wk.shape)} "
                         f"bq {tuple(bq.shape)} bk {tuple(bk.shape)}")
    if not (wq.dtype == wk.dtype == bq.dtype == bk.dtype):
        raise TypeError("scorer parameters must share one dtype")
    return N.Scorer(wq.data_ptr(), bq.data_ptr(), wk.data_ptr(), bk.data_ptr(), d, hd, _code(wq))


_seg_cache: dict = {}


def _uniform_segments(b: int, n: int, k: int) -> N.Segments:
    return N.Segments(b, n, b * n, k, b * k, None, None)


def _ragged_segments(seg_lens: Sequence[int], ks: Sequence[int], device) -> Tuple[N.Segments, torch.Tensor, torch.Tensor]:
    key = (tuple(seg_lens), tuple(ks), str(device))
    hit = _seg_cache.get(key)
    if hit is None:
        cu_r = torch.tensor([0] + list(seg_lens), dtype=torch.int64).cumsum(0).to(torch.int32)
        cu_o = torch.tensor([0] + list(ks), dtype=torch.int64).cumsum(0).to(torch.int32)
        hit = (cu_r.to(device), cu_o.to(device), int(cu_r[-1]), int(cu_o[-1]))
        if len(_seg_cache) > 256:
            _seg_cache.clear()
        _seg_cache[key] = hit
    cu_r, cu_o, total, total_out = hit
    seg = N.Segments(len(seg_lens), max(seg_lens), total, max(ks), total_out, cu_r.data_ptr(), cu_o.data_ptr())
    return seg, cu_r, cu_o


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)


def _as_bnd(h: torch.Tensor) -> Tuple[int, int, int]:
    if h.dim() == 2:
        return 1, h.shape[0], h.shape[1]
    if h.dim() == 3:
        return h.shape[0], h.shape[1], h.shape[2]
    raise ValueError(f"tokens must be [N, D] or [B, N, D], got {tuple(h.shape)}")


# ------------------------------------------------------------------------------------------------
# inference
# ------------------------------------------------------------------------------------------------

def lis_scores(h, wq, bq, wk, bk) -> torch.Tensor:
    """TransformerScorer.forward: h [B,N,D] (or [N,D]) -> fp32 scores [B,N] (or [N])."""
    dev = _dev(h, wq, bq, wk, bk)
    b, n, d = _as_bnd(h)
    sc = _scorer(wq, bq, wk, bk)
    if sc.d != d:
        raise ValueError(f"token width {d} != scorer in_features {sc.d}")
    seg = _uniform_segments(b, n, 1)
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_workspace_bytes(C.byref(seg), d, sc.hd), dev)
    scores = torch.empty(h.shape[:-1], dtype=torch.float32, device=dev)
    N.check(lib.vsel_lis_scores(_stream(), h.data_ptr(), _code(h), C.byref(seg), C.byref(sc), ws.data_ptr(), ws.numel(),
                                scores.data_ptr()))
    return scores


def lis_select(h, wq, bq, wk, bk, k: int):
    """Fused score + hard top-k + gather.  h [B,N,D] or [N,D] -> (out [B,k,D], idx int64 [B,k] ascending, scores fp32 [B,N])."""
    dev = _dev(h, wq, bq, wk, bk)
    b, n, d = _as_bnd(h)
    sc = _scorer(wq, bq, wk, bk)
    if sc.d != d:
        raise ValueError(f"token width {d} != scorer in_features {sc.d}")
    seg = _uniform_segments(b, n, int(k))
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_workspace_bytes(C.byref(seg), d, sc.hd), dev)
    lead = h.shape[:-2]
    out = torch.empty(*lead, k, d, dtype=h.dtype, device=dev)
    idx = torch.empty(*lead, k, dtype=torch.int64, device=dev)
    scores = torch.empty(*lead, n, dtype=torch.float32, device=dev)
    N.check(lib.vsel_lis_select(_stream(), h.data_ptr(), _code(h), C.byref(seg), C.byref(sc), ws.data_ptr(), ws.numel(),
                                out.data_ptr(), idx.data_ptr(), scores.data_ptr()))
    return out, idx, scores


def lis_select_permuted(h_physical, logical_to_physical, physical_to_logical, wq, bq, wk, bk, k: int):
    """lis_select on h_physical[logical_to_physical] without materialising that gather (Qwen2.5-VL window order ->
    natural order: logical_to_physical = reverse_indices, physical_to_logical = window_index).  h [N,D] or [B,N,D];
    maps int64 with GLOBAL row numbers."""
    dev = _dev(h_physical, logical_to_physical, physical_to_logical, wq, bq, wk, bk)
    b, n, d = _as_bnd(h_physical)
    if logical_to_physical.dtype != torch.int64 or physical_to_logical.dtype != torch.int64:
        raise TypeError("permutation maps must be int64")
    if logical_to_physical.numel() != b * n or physical_to_logical.numel() != b * n:
        raise ValueError("permutation maps must have one entry per token row")
    sc = _scorer(wq, bq, wk, bk)
    seg = _uniform_segments(b, n, int(k))
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_workspace_bytes(C.byref(seg), d, sc.hd), dev)
    lead = h_physical.shape[:-2]
    out = torch.empty(*lead, k, d, dtype=h_physical.dtype, device=dev)
    idx = torch.empty(*lead, k, dtype=torch.int64, device=dev)
    scores = torch.empty(*lead, n, dtype=torch.float32, device=dev)
    N.check(lib.vsel_lis_select_permuted(_stream(), h_physical.data_ptr(), _code(h_physical), C.byref(seg), C.byref(sc),
                                         ws.data_ptr(), ws.numel(), logical_to_physical.data_ptr(),
                                         physical_to_logical.data_ptr(), out.data_ptr(), idx.data_ptr(), scores.data_ptr()))
    return out, idx, scores


def gelu_colsum(x: torch.Tensor, n_seg: int = 1, sums: bool = True):
    """y = GELU(x) (erf form, as nn.GELU()) and the per-segment column sums of y, in one pass.  x [R, C] (n_seg equal
    segments of R / n_seg rows) -> (y [R, C] same dtype, col_sums fp32 [n_seg, C]).  sums=False: the same kernel without the
    sums -> (y, None)."""
    dev = _dev(x)
    if x.dim() != 2 or not x.is_contiguous():
        raise ValueError("gelu_colsum takes a contiguous [rows, cols] tensor")
    r, c = x.shape
    if n_seg < 1 or r % n_seg:
        raise ValueError(f"rows {r} do not split into {n_seg} equal segments")
    seg = _uniform_segments(n_seg, r // n_seg, 1)
    lib = N.lib()
    y = torch.empty_like(x)
    if not sums:
        N.check(lib.vsel_gelu_colsum(_stream(), x.data_ptr(), _code(x), C.byref(seg), c, y.data_ptr(), None, None, 0))
        return y, None
    ws = _workspace(lib.vsel_gelu_colsum_workspace_bytes(C.byref(seg), c), dev)
    col_sums = torch.empty(n_seg, c, dtype=torch.float32, device=dev)
    N.check(lib.vsel_gelu_colsum(_stream(), x.data_ptr(), _code(x), C.byref(seg), c, y.data_ptr(), col_sums.data_ptr(),
                                 ws.data_ptr(), ws.numel()))
    return y, col_sums


def colsum_linear(col_sums_in: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], rows_per_seg: int):
    """sum_rows(H) from sum_rows(G) for H = G W^T + b on the stored weight (vsel_colsum_linear): col_sums_in fp32 [S, Cin],
    weight [Cout, Cin] (bf16 / fp32), bias [Cout] or None, rows_per_seg = N -> fp32 [S, Cout] = col_sums_in W^T + N b."""
    dev = _dev(col_sums_in, weight, bias)
    if col_sums_in.dtype != torch.float32 or col_sums_in.dim() != 2 or not col_sums_in.is_contiguous():
        raise ValueError("col_sums_in must be contiguous float32 [n_seg, Cin]")
    if weight.dim() != 2 or weight.shape[1] != col_sums_in.shape[1] or not weight.is_contiguous():
        raise ValueError("weight must be contiguous [Cout, Cin]")
    if bias is not None and (bias.dtype != weight.dtype or bias.numel() != weight.shape[0]):
        raise ValueError("bias must be [Cout] of the weight's dtype")
    s, cin = col_sums_in.shape
    cout = weight.shape[0]
    seg = _uniform_segments(s, int(rows_per_seg), 1)
    lib = N.lib()
    ws = _workspace(lib.vsel_colsum_linear_workspace_bytes(s, cin, cout), dev)
    out = torch.empty(s, cout, dtype=torch.float32, device=dev)
    b = None if bias is None else bias.contiguous()
    N.check(lib.vsel_colsum_linear(_stream(), col_sums_in.data_ptr(), C.byref(seg), weight.data_ptr(), _p(b), _code(weight), cin, cout,
                                   out.data_ptr(), ws.data_ptr(), ws.numel()))
    return out


def lis_select_presummed(h, col_sums, wq, bq, wk, bk, k: int, logical_to_physical=None, physical_to_logical=None):
    """lis_select / lis_select_permuted with the column sums of the tokens supplied by their producer (fp32 [B, D] or [D]):
    the first HBM sweep over h is skipped.  h [N,D] or [B,N,D]."""
    dev = _dev(h, col_sums, wq, bq, wk, bk, logical_to_physical, physical_to_logical)
    b, n, d = _as_bnd(h)
    if col_sums.dtype != torch.float32 or col_sums.numel() != b * d or not col_sums.is_contiguous():
        raise ValueError("col_sums must be contiguous float32 with one row of D sums per segment")
    if (logical_to_physical is None) != (physical_to_logical is None):
        raise ValueError("give both permutation maps or neither")
    if logical_to_physical is not None:
        if logical_to_physical.dtype != torch.int64 or physical_to_logical.dtype != torch.int64:
            raise TypeError("permutation maps must be int64")
        if logical_to_physical.numel() != b * n or physical_to_logical.numel() != b * n:
            raise ValueError("permutation maps must have one entry per token row")
    sc = _scorer(wq, bq, wk, bk)
    seg = _uniform_segments(b, n, int(k))
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_workspace_bytes(C.byref(seg), d, sc.hd), dev)
    lead = h.shape[:-2]
    out = torch.empty(*lead, k, d, dtype=h.dtype, device=dev)
    idx = torch.empty(*lead, k, dtype=torch.int64, device=dev)
    scores = torch.empty(*lead, n, dtype=torch.float32, device=dev)
    N.check(lib.vsel_lis_select_presummed(_stream(), h.data_ptr(), _code(h), C.byref(seg), C.byref(sc), col_sums.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _p(logical_to_physical), _p(physical_to_logical),
                                          out.data_ptr(), idx.data_ptr(), scores.data_ptr()))
    return out, idx, scores


def lis_select_varlen(h, seg_lens: Sequence[int], ks: Sequence[int], wq, bq, wk, bk):
    """Ragged form: h [T,D] holds len(seg_lens) segments back to back; segment s keeps ks[s] rows.
    -> (out [sum ks, D], idx int64 [sum ks] local to the segment, scores fp32 [T])."""
    dev = _dev(h, wq, bq, wk, bk)
    if h.dim() != 2:
        raise ValueError("ragged tokens must be [T, D]")
    if len(seg_lens) != len(ks) or len(ks) == 0:
        raise ValueError("seg_lens and ks must have the same non-zero length")
    for n_s, k_s in zip(seg_lens, ks):
        if not (1 <= k_s <= n_s):
            raise ValueError(f"need 1 <= k <= rows per segment, got k={k_s}, rows={n_s}")
    t, d = h.shape
    if sum(seg_lens) != t:
        raise ValueError(f"sum(seg_lens)={sum(seg_lens)} != rows {t}")
    sc = _scorer(wq, bq, wk, bk)
    seg, cu_r, cu_o = _ragged_segments(seg_lens, ks, dev)
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_workspace_bytes(C.byref(seg), d, sc.hd), dev)
    out = torch.empty(seg.total_out, d, dtype=h.dtype, device=dev)
    idx = torch.empty(seg.total_out, dtype=torch.int64, device=dev)
    scores = torch.empty(t, dtype=torch.float32, device=dev)
    N.check(lib.vsel_lis_select(_stream(), h.data_ptr(), _code(h), C.byref(seg), C.byref(sc), ws.data_ptr(), ws.numel(),
                                out.data_ptr(), idx.data_ptr(), scores.data_ptr()))
    return out, idx, scores


def hard_topk(scores: torch.Tensor, k: int, want_mask: bool = False):
    """scores fp32 [B,N] or [N] -> idx int64 [B,k] ascending (topk(k).indices.sort()); optional 0/1 mask."""
    dev = _dev(scores)
    if scores.dtype != torch.float32:
        raise TypeError("hard_topk takes float32 scores")
    b = 1 if scores.dim() == 1 else scores.shape[0]
    n = scores.shape[-1]
    seg = _uniform_segments(b, n, int(k))
    idx = torch.empty(*scores.shape[:-1], k, dtype=torch.int64, device=dev)
    mask = torch.empty_like(scores) if want_mask else None
    N.check(N.lib().vsel_topk_select(_stream(), scores.data_ptr(), C.byref(seg), idx.data_ptr(), _p(mask)))
    return (idx, mask) if want_mask else idx


def gather_rows(h: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """h [B,N,D] / [N,D], idx int64 [B,k] / [k] -> h[idx] per segment."""
    dev = _dev(h, idx)
    b, n, d = _as_bnd(h)
    k = idx.shape[-1]
    seg = _uniform_segments(b, n, k)
    out = torch.empty(*h.shape[:-2], k, d, dtype=h.dtype, device=dev)
    N.check(N.lib().vsel_gather_rows(_stream(), h.data_ptr(), _code(h), d, C.byref(seg), idx.data_ptr(), out.data_ptr()))
    return out


# ------------------------------------------------------------------------------------------------
# differentiable top-k
# ------------------------------------------------------------------------------------------------

def soft_topk_fwd(xs: torch.Tensor, k: int, bf16_reference: bool = False):
    """_find_ts: (ps [B, N], ts [B]).  bf16_reference: the reference's own bfloat16 arithmetic (every step rounded to bf16: its
    bisection stalls on a bf16 neighbour of the root) instead of the fp32 root -- vsel_soft_topk_fwd_bf16ref, include/vsel.h."""
    dev = _dev(xs)
    if xs.dtype != torch.float32 or xs.dim() != 2:
        raise TypeError("soft_topk_fwd takes float32 [B, N]")
    b, n = xs.shape
    ps = torch.empty_like(xs)
    ts = torch.empty(b, dtype=torch.float32, device=dev)
    fn = N.lib().vsel_soft_topk_fwd_bf16ref if bf16_reference else N.lib().vsel_soft_topk_fwd
    N.check(fn(_stream(), xs.data_ptr(), b, n, int(k), ps.data_ptr(), ts.data_ptr()))
    return ps, ts


def soft_topk_bwd(grad_ps: torch.Tensor, xs: torch.Tensor, ts: torch.Tensor) -> torch.Tensor:
    _dev(grad_ps, xs, ts)
    b, n = xs.shape
    out = torch.empty_like(xs)
    N.check(N.lib().vsel_soft_topk_bwd(_stream(), grad_ps.data_ptr(), xs.data_ptr(), ts.data_ptr(), b, n, out.data_ptr()))
    return out


# ------------------------------------------------------------------------------------------------
# training block
# ------------------------------------------------------------------------------------------------

def lis_train_fwd(h, wq, bq, wk, bk, k: int):
    """One segment h [N,D] -> (h_new [N,D], ps [N], y [N], scores [N], ts [1], bce [1])."""
    dev = _dev(h, wq, bq, wk, bk)
    n, d = h.shape
    sc = _scorer(wq, bq, wk, bk)
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_train_workspace_bytes(n, d, sc.hd), dev)
    h_new = torch.empty_like(h)
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    ps, y, scores, ts, bce = f(n), f(n), f(n), f(1), f(1)
    N.check(lib.vsel_lis_train_fwd(_stream(), h.data_ptr(), _code(h), n, int(k), C.byref(sc), ws.data_ptr(), ws.numel(),
                                   h_new.data_ptr(), ps.data_ptr(), y.data_ptr(), scores.data_ptr(), ts.data_ptr(),
                                   bce.data_ptr()))
    return h_new, ps, y, scores, ts, bce


def lis_train_bwd(d_hnew, h, wq, bq, wk, bk, ps, y, scores, ts, d_ps_ext=None, dl_dbce: float = 0.0, need_dh: bool = False,
                  out=None):
    """-> (dwq, dbq, dwk, dbk fp32, dh or None).  out = (dwq, dbq, dwk, dbk): contiguous fp32 buffers to OVERWRITE (e.g. the
    slices of a flat all-reduce bucket, ddp.LisGradSync.views()) instead of fresh tensors."""
    dev = _dev(d_hnew, h, wq, bq, wk, bk, ps, y, scores, ts, d_ps_ext)
    n, d = h.shape
    sc = _scorer(wq, bq, wk, bk)
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_train_workspace_bytes(n, d, sc.hd), dev)
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    if out is not None:
        dwq, dbq, dwk, dbk = out
        for t, shape in ((dwq, (sc.hd, d)), (dbq, (sc.hd,)), (dwk, (sc.hd, d)), (dbk, (sc.hd,))):
            if t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous() or t.device != dev:
                raise ValueError("lis_train_bwd(out=...): need contiguous float32 [Hd,D], [Hd], [Hd,D], [Hd] on the same device")
    else:
        dwq, dbq, dwk, dbk = f(sc.hd, d), f(sc.hd), f(sc.hd, d), f(sc.hd)
    dh = torch.empty_like(h) if need_dh else None
    N.check(lib.vsel_lis_train_bwd(_stream(), d_hnew.data_ptr(), h.data_ptr(), _code(h), n, C.byref(sc), ps.data_ptr(),
                                   y.data_ptr(), scores.data_ptr(), ts.data_ptr(), _p(d_ps_ext), float(dl_dbce),
                                   ws.data_ptr(), ws.numel(), dwq.data_ptr(), dbq.data_ptr(), dwk.data_ptr(),
                                   dbk.data_ptr(), _p(dh)))
    return dwq, dbq, dwk, dbk, dh


def lis_train_bwd_factors(d_hnew, h, wq, bq, wk, bk, ps, y, scores, ts, d_ps_ext=None, dl_dbce: float = 0.0,
                          need_dh: bool = False, out=None):
    """The same backward with the weight gradients as rank-1 factors: -> (payload fp32 [2 (Hd + D) + 2 Hd], dh or None),
    payload = a [Hd] | gx [D] | dk [Hd] | xsum [D] | dbq [Hd] | dbk [Hd]  with  dWq = a (x) gx,  dWk = dk (x) xsum
    (factor_payload_split / ddp.LisFactorSync turn payloads into dense gradients).  out = a contiguous fp32 row to overwrite."""
    dev = _dev(d_hnew, h, wq, bq, wk, bk, ps, y, scores, ts, d_ps_ext)
    n, d = h.shape
    sc = _scorer(wq, bq, wk, bk)
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_train_workspace_bytes(n, d, sc.hd), dev)
    numel = 2 * (sc.hd + d) + 2 * sc.hd
    if out is None:
        out = torch.empty(numel, dtype=torch.float32, device=dev)
    elif out.dtype != torch.float32 or out.numel() != numel or not out.is_contiguous() or out.device != dev:
        raise ValueError(f"lis_train_bwd_factors(out=...): need a contiguous float32 buffer of {numel} elements on the same device")
    dh = torch.empty_like(h) if need_dh else None
    base = out.data_ptr()
    fac_bytes = 2 * (sc.hd + d) * 4
    N.check(lib.vsel_lis_train_bwd_factors(_stream(), d_hnew.data_ptr(), h.data_ptr(), _code(h), n, C.byref(sc), ps.data_ptr(),
                                           y.data_ptr(), scores.data_ptr(), ts.data_ptr(), _p(d_ps_ext), float(dl_dbce),
                                           ws.data_ptr(), ws.numel(), base, base + fac_bytes, base + fac_bytes + 4 * sc.hd,
                                           _p(dh)))
    return out, dh


def factors_to_grads(payload: torch.Tensor, hd: int, d: int, scale: float = 1.0, out=None):
    """payload fp32 [R, 2 (Hd + D) + 2 Hd] (rows of lis_train_bwd_factors, own and / or all-gathered) ->
    (dwq [Hd, D], dbq [Hd], dwk [Hd, D], dbk [Hd]) = scale x the sums over the rows, in row order."""
    dev = _dev(payload)
    row = 2 * (hd + d) + 2 * hd
    if payload.dtype != torch.float32 or payload.dim() != 2 or payload.shape[1] != row or not payload.is_contiguous():
        raise ValueError(f"payload must be contiguous float32 [R, {row}]")
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    dwq, dbq, dwk, dbk = out if out is not None else (f(hd, d), f(hd), f(hd, d), f(hd))
    N.check(N.lib().vsel_lis_factors_to_grads(_stream(), payload.data_ptr(), payload.shape[0], hd, d, float(scale), dwq.data_ptr(),
                                              dbq.data_ptr(), dwk.data_ptr(), dbk.data_ptr()))
    return dwq, dbq, dwk, dbk


def factor_payload_split(payload: torch.Tensor, hd: int, d: int):
    """payload [..., 2 (Hd + D) + 2 Hd] -> (a [..., Hd], gx [..., D], dk [..., Hd], xsum [..., D], dbq [..., Hd], dbk [..., Hd])."""
    return torch.split(payload, [hd, d, hd, d, hd, hd], dim=-1)


def lis_scores_bwd(g, h, wq, bq, wk, bk, need_dh: bool = False):
    """Backward of lis_scores for one segment: g fp32 [N], h [N,D] -> (dwq, dbq, dwk, dbk fp32, dh or None)."""
    dev = _dev(g, h, wq, bq, wk, bk)
    n, d = h.shape
    sc = _scorer(wq, bq, wk, bk)
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_train_workspace_bytes(n, d, sc.hd), dev)
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    dwq, dbq, dwk, dbk = f(sc.hd, d), f(sc.hd), f(sc.hd, d), f(sc.hd)
    dh = torch.empty_like(h) if need_dh else None
    N.check(lib.vsel_lis_scores_bwd(_stream(), g.data_ptr(), h.data_ptr(), _code(h), n, C.byref(sc), ws.data_ptr(),
                                    ws.numel(), dwq.data_ptr(), dbq.data_ptr(), dwk.data_ptr(), dbk.data_ptr(), _p(dh)))
    return dwq, dbq, dwk, dbk, dh


# ------------------------------------------------------------------------------------------------
# sequence splice
# ------------------------------------------------------------------------------------------------

def splice(input_ids, inputs_embeds, visual_token_id: int, all_indices, visual_embeds, n_visual: int,
           position_ids=None, attention_mask=None, check: bool = False):
    """Batch-1 splice on device.  input_ids [1, L] int64, inputs_embeds [1, L, D], all_indices [k] int64 ascending,
    visual_embeds [k, D], position_ids [R, 1, L] int64 or None, attention_mask [1, L] or None ->
    (selected_indices [L'], input_ids' [1, L'], inputs_embeds' [1, L', D], position_ids' [R, 1, L'] | None, attention_mask' | None).
    check=True synchronises and raises ValueError on a token-count mismatch (reference: selector_model.py:210-213)."""
    dev = _dev(input_ids, inputs_embeds, all_indices, visual_embeds, position_ids, attention_mask)
    if input_ids.dim() != 2 or input_ids.shape[0] != 1:
        raise ValueError("selector only support single batch")              # reference assert, EV :270
    if input_ids.dtype != torch.int64 or all_indices.dtype != torch.int64:
        raise TypeError("input_ids / all_indices must be int64")
    L = input_ids.shape[1]
    d = inputs_embeds.shape[-1]
    k = all_indices.numel()
    l_out = L - int(n_visual) + k
    vis = visual_embeds.to(inputs_embeds.dtype).contiguous()
    am = None
    if attention_mask is not None:
        am = attention_mask.to(torch.int64).contiguous()
    pos = None
    rows = 0
    if position_ids is not None:
        pos = position_ids.to(torch.int64).contiguous()
        rows = pos.numel() // L
    sel = torch.empty(l_out, dtype=torch.int64, device=dev)
    new_ids = torch.empty(1, l_out, dtype=torch.int64, device=dev)
    new_emb = torch.empty(1, l_out, d, dtype=inputs_embeds.dtype, device=dev)
    new_pos = torch.empty(rows, 1, l_out, dtype=torch.int64, device=dev) if pos is not None else None
    new_am = torch.empty(1, l_out, dtype=torch.int64, device=dev) if am is not None else None
    src = torch.empty(max(l_out, 1), dtype=torch.int32, device=dev)
    stats = torch.empty(3, dtype=torch.int32, device=dev)
    if l_out == 0:          # every position was a dropped visual token: nothing to write
        return sel, new_ids, new_emb, new_pos, new_am
    N.check(N.lib().vsel_splice(_stream(), input_ids.data_ptr(), L, int(visual_token_id), all_indices.data_ptr(), k,
                                int(n_visual), inputs_embeds.data_ptr(), vis.data_ptr(), _code(inputs_embeds), d, _p(pos), rows,
                                _p(am), sel.data_ptr(), new_ids.data_ptr(), new_emb.data_ptr(), _p(new_pos), _p(new_am),
                                src.data_ptr(), stats.data_ptr()))
    if check:
        found, written, kept = stats.tolist()
        if found != n_visual or written != l_out or kept != k:
            raise ValueError(f"Image features and image tokens do not match: tokens: {found}, features {n_visual}")
    if new_am is not None and attention_mask.dtype != torch.int64:
        new_am = new_am.to(attention_mask.dtype)
    return sel, new_ids, new_emb, new_pos, new_am


def splice_batched(input_ids, inputs_embeds, visual_token_id: int, seq_lens: Sequence[int], visual_lens: Sequence[int],
                   ks: Sequence[int], all_indices, visual_embeds, position_ids=None, check: bool = False):
    """Packed-batch splice.  input_ids [T] int64 = S prompts back to back (seq_lens), prompt s holding visual_lens[s]
    visual tokens of which ks[s] are kept; all_indices [sum ks] int64 = local ranks per prompt, ascending (the idx output of
    lis_select_varlen(h, visual_lens, ks, ...)); visual_embeds [sum ks, D] (its `out`); inputs_embeds [T, D];
    position_ids [R, T] int64 or None ->
    (selected_indices [T'], input_ids' [T'], inputs_embeds' [T', D], position_ids' [R, T'] | None, cu_seqlens' int32 [S+1]).
    Each prompt is spliced exactly like the reference's batch-1 forward; no host sync unless check=True (raises ValueError
    on a token-count mismatch, reference: selector_model.py:210-213)."""
    dev = _dev(input_ids, inputs_embeds, all_indices, visual_embeds, position_ids)
    if input_ids.dim() != 1 or inputs_embeds.dim() != 2:
        raise ValueError("packed splice takes input_ids [T] and inputs_embeds [T, D]")
    if input_ids.dtype != torch.int64 or all_indices.dtype != torch.int64:
        raise TypeError("input_ids / all_indices must be int64")
    s = len(seq_lens)
    if s == 0 or len(visual_lens) != s or len(ks) != s:
        raise ValueError("seq_lens, visual_lens and ks must have the same non-zero length")
    for l_s, n_s, k_s in zip(seq_lens, visual_lens, ks):
        if not (0 <= k_s <= n_s <= l_s):
            raise ValueError(f"need 0 <= k <= visual tokens <= length per prompt, got k={k_s}, visual={n_s}, length={l_s}")
    t = input_ids.numel()
    if sum(seq_lens) != t or inputs_embeds.shape[0] != t:
        raise ValueError(f"sum(seq_lens)={sum(seq_lens)} / embeds rows {inputs_embeds.shape[0]} != positions {t}")
    n_tot, k_tot = sum(visual_lens), sum(ks)
    if all_indices.numel() != k_tot or visual_embeds.shape[0] != k_tot:
        raise ValueError("all_indices / visual_embeds must hold sum(ks) rows")
    d = inputs_embeds.shape[-1]
    l_out = t - n_tot + k_tot

    def _cu(lens):
        c = [0]
        for x in lens:
            c.append(c[-1] + int(x))
        return torch.tensor(c, dtype=torch.int32).to(dev, non_blocking=True)

    cu_s, cu_v, cu_k = _cu(seq_lens), _cu(visual_lens), _cu(ks)
    vis = visual_embeds.to(inputs_embeds.dtype).contiguous()
    emb = inputs_embeds.contiguous()
    pos = None
    rows = 0
    if position_ids is not None:
        pos = position_ids.to(torch.int64).contiguous()
        rows = pos.numel() // t
    sel = torch.empty(l_out, dtype=torch.int64, device=dev)
    new_ids = torch.empty(l_out, dtype=torch.int64, device=dev)
    new_emb = torch.empty(l_out, d, dtype=inputs_embeds.dtype, device=dev)
    new_pos = torch.empty(rows, l_out, dtype=torch.int64, device=dev) if pos is not None else None
    cu_out = torch.empty(s + 1, dtype=torch.int32, device=dev)
    src = torch.empty(max(l_out, 1), dtype=torch.int32, device=dev)
    stats = torch.empty(4, dtype=torch.int32, device=dev)
    N.check(N.lib().vsel_splice_batched(_stream(), input_ids.data_ptr(), t, cu_s.data_ptr(), cu_v.data_ptr(), cu_k.data_ptr(),
                                        s, max(visual_lens), n_tot, k_tot, int(visual_token_id), _p(all_indices), emb.data_ptr(),
                                        _p(vis), _code(inputs_embeds), d, _p(pos), rows, sel.data_ptr(), new_ids.data_ptr(),
                                        new_emb.data_ptr(), _p(new_pos), cu_out.data_ptr(), src.data_ptr(), stats.data_ptr()))
    if check:
        found, written, kept, bad = stats.tolist()
        if bad or found != n_tot or written != l_out or kept != k_tot:
            raise ValueError(f"Image features and image tokens do not match: tokens: {found}, features {n_tot}")
    return sel, new_ids, new_emb, new_pos, cu_out


def _select_splice_common(h, input_ids, inputs_embeds, seq_lens, visual_lens, ks, position_ids, attention_mask, dev):
    """Shared shape checks / output allocation of lis_select_splice and topk_select_splice."""
    if h.dim() != 2 or inputs_embeds.dim() != 2 or input_ids.dim() != 1:
        raise ValueError("select-splice takes h [sum N, D], input_ids [T], inputs_embeds [T, D]")
    if input_ids.dtype != torch.int64:
        raise TypeError("input_ids must be int64")
    if h.dtype != inputs_embeds.dtype or h.shape[1] != inputs_embeds.shape[1]:
        raise ValueError("fused select-splice needs tokens and embeddings of the same dtype and width "
                         f"(got {h.dtype} x {h.shape[1]} and {inputs_embeds.dtype} x {inputs_embeds.shape[1]}); use lis_select + splice")
    s = len(seq_lens)
    if s == 0 or len(visual_lens) != s or len(ks) != s:
        raise ValueError("seq_lens, visual_lens and ks must have the same non-zero length")
    for l_s, n_s, k_s in zip(seq_lens, visual_lens, ks):
        if not (1 <= k_s <= n_s <= l_s):
            raise ValueError(f"need 1 <= k <= visual tokens <= length per prompt, got k={k_s}, visual={n_s}, length={l_s}")
    t = input_ids.numel()
    if sum(seq_lens) != t or inputs_embeds.shape[0] != t or sum(visual_lens) != h.shape[0]:
        raise ValueError("sum(seq_lens) / embeds rows / sum(visual_lens) do not match the tensors")
    if s == 1:
        seg, cu_r, cu_o, cu_s = _uniform_segments(1, int(visual_lens[0]), int(ks[0])), None, None, None
    else:
        seg, cu_r, cu_o = _ragged_segments(visual_lens, ks, dev)
        c = [0]
        for x in seq_lens:
            c.append(c[-1] + int(x))
        cu_s = torch.tensor(c, dtype=torch.int32).to(dev, non_blocking=True)
    n_tot, k_tot = sum(visual_lens), sum(ks)
    l_out = t - n_tot + k_tot
    d = h.shape[1]
    pos, rows = None, 0
    if position_ids is not None:
        pos = position_ids.to(torch.int64).contiguous()
        rows = pos.numel() // t
    am = None if attention_mask is None else attention_mask.to(torch.int64).contiguous()
    out = dict(idx=torch.empty(k_tot, dtype=torch.int64, device=dev), sel=torch.empty(l_out, dtype=torch.int64, device=dev),
               new_ids=torch.empty(l_out, dtype=torch.int64, device=dev),
               new_emb=torch.empty(l_out, d, dtype=inputs_embeds.dtype, device=dev),
               new_pos=torch.empty(rows, l_out, dtype=torch.int64, device=dev) if pos is not None else None,
               new_am=torch.empty(l_out, dtype=torch.int64, device=dev) if am is not None else None,
               cu_out=torch.empty(s + 1, dtype=torch.int32, device=dev), src=torch.empty(l_out, dtype=torch.int32, device=dev),
               stats=torch.empty(4, dtype=torch.int32, device=dev))
    max_len_out = max(l - n + k for l, n, k in zip(seq_lens, visual_lens, ks))
    return seg, (cu_r, cu_o), cu_s, pos, rows, am, out, l_out, max_len_out, n_tot, k_tot


def _row_map(m, name: str, rows: int):
    """A row permutation map as the C-ABI reads it: int64, contiguous, one entry per token row (or None)."""
    if m is None:
        return None
    if m.dtype != torch.int64:
        raise TypeError(f"{name} must be int64 (got {m.dtype})")
    if m.numel() != rows:
        raise ValueError(f"{name} must have one entry per token row ({rows}), got {m.numel()}")
    return m.contiguous()


def _select_splice_finish(o, check, n_tot, l_out, k_tot, attention_mask):
    if check:
        found, written, kept, bad = o["stats"].tolist()
        if bad or found != n_tot or written != l_out or kept != k_tot:
            raise ValueError(f"Image features and image tokens do not match: tokens: {found}, features {n_tot}")
    new_am = o["new_am"]
    if new_am is not None and attention_mask.dtype != torch.int64:
        new_am = new_am.to(attention_mask.dtype)
    return new_am


def lis_select_splice(h, wq, bq, wk, bk, input_ids, inputs_embeds, visual_token_id: int, seq_lens: Sequence[int],
                      visual_lens: Sequence[int], ks: Sequence[int], position_ids=None, attention_mask=None, col_sums=None,
                      logical_to_physical=None, physical_to_logical=None, check: bool = False, soft: bool = False):
    """Scores + hard top-k + splice with the kept rows written once, from the token tensor into inputs_embeds'
    (vsel_lis_select_splice).  h [sum N, D] = the visual tokens of S prompts back to back (prompt s: visual_lens[s] rows, ONE
    jointly scored segment, ks[s] kept); input_ids [T] / inputs_embeds [T, D] = the prompts back to back (seq_lens);
    position_ids [R, T] or None; attention_mask [T] or None ->
    dict(idx [sum k] local ranks ascending, scores fp32 [sum N], selected_indices [T'], input_ids [T'], inputs_embeds [T', D],
         position_ids [R, T'] | None, attention_mask [T'] | None, cu_seqlens int32 [S+1]).
    Bit-identical to lis_select(_varlen / _permuted / _presummed) followed by splice(_batched).
    soft=True (uniform visual_lens / ks): also soft_ps fp32 [sum N], soft_ts fp32 [S] = soft_topk_fwd(scores, k) bit for bit (the
    reference's last_combined_scores, EV :190); for one prompt of <= 4096 tokens without a launch of its own.  None when not
    0 < k < N."""
    dev = _dev(h, wq, bq, wk, bk, input_ids, inputs_embeds, position_ids, attention_mask, col_sums, logical_to_physical,
               physical_to_logical)
    seg, _keep, cu_s, pos, rows, am, o, l_out, max_len_out, n_tot, k_tot = _select_splice_common(
        h, input_ids, inputs_embeds, seq_lens, visual_lens, ks, position_ids, attention_mask, dev)
    sc = _scorer(wq, bq, wk, bk)
    if sc.d != h.shape[1]:
        raise ValueError(f"token width {h.shape[1]} != scorer in_features {sc.d}")
    if (logical_to_physical is None) != (physical_to_logical is None):
        raise ValueError("give both permutation maps or neither")
    if col_sums is not None and (col_sums.dtype != torch.float32 or col_sums.numel() != len(seq_lens) * sc.d):
        raise ValueError("col_sums must be float32 with one row of D sums per prompt")
    col_sums = None if col_sums is None else col_sums.contiguous()
    logical_to_physical = _row_map(logical_to_physical, "logical_to_physical", n_tot)
    physical_to_logical = _row_map(physical_to_logical, "physical_to_logical", n_tot)
    ids = input_ids.contiguous()
    lib = N.lib()
    ws = _workspace(lib.vsel_lis_workspace_bytes(C.byref(seg), sc.d, sc.hd), dev)
    scores = torch.empty(n_tot, dtype=torch.float32, device=dev)
    hc, emb = h.contiguous(), inputs_embeds.contiguous()
    soft_ps = soft_ts = None
    if soft:
        if len(set(visual_lens)) != 1 or len(set(ks)) != 1:
            raise ValueError("soft=True needs uniform segments (the reference's call is one prompt)")
        if 0 < ks[0] < visual_lens[0]:
            both = torch.empty(n_tot + len(seq_lens), dtype=torch.float32, device=dev)      # one allocation: ps | ts
            soft_ps, soft_ts = both[:n_tot], both[n_tot:]
    N.check(lib.vsel_lis_select_splice(
        _stream(), hc.data_ptr(), _code(hc), C.byref(seg), C.byref(sc), ws.data_ptr(), ws.numel(), _p(col_sums),
        _p(logical_to_physical), _p(physical_to_logical), ids.data_ptr(), ids.numel(), _p(cu_s), max_len_out,
        int(visual_token_id), emb.data_ptr(), _p(pos), rows, _p(am), o["idx"].data_ptr(), scores.data_ptr(), o["sel"].data_ptr(),
        o["new_ids"].data_ptr(), o["new_emb"].data_ptr(), _p(o["new_pos"]), _p(o["new_am"]), o["cu_out"].data_ptr(),
        o["src"].data_ptr(), o["stats"].data_ptr(), _p(soft_ps), _p(soft_ts)))
    new_am = _select_splice_finish(o, check, n_tot, l_out, k_tot, attention_mask)
    return dict(idx=o["idx"], scores=scores, selected_indices=o["sel"], input_ids=o["new_ids"], inputs_embeds=o["new_emb"],
                position_ids=o["new_pos"], attention_mask=new_am, cu_seqlens=o["cu_out"], soft_ps=soft_ps, soft_ts=soft_ts)


def topk_select_splice(scores, h, input_ids, inputs_embeds, visual_token_id: int, seq_lens: Sequence[int],
                       visual_lens: Sequence[int], ks: Sequence[int], position_ids=None, attention_mask=None,
                       logical_to_physical=None, check: bool = False):
    """The same on given fp32 scores [sum N] (vsel_topk_select_splice): hard top-k + splice, kept rows read from h."""
    dev = _dev(scores, h, input_ids, inputs_embeds, position_ids, attention_mask, logical_to_physical)
    if scores.dtype != torch.float32 or scores.numel() != h.shape[0]:
        raise TypeError("scores must be float32 with one entry per token row")
    seg, _keep, cu_s, pos, rows, am, o, l_out, max_len_out, n_tot, k_tot = _select_splice_common(
        h, input_ids, inputs_embeds, seq_lens, visual_lens, ks, position_ids, attention_mask, dev)
    hc, emb, scc, ids = h.contiguous(), inputs_embeds.contiguous(), scores.contiguous(), input_ids.contiguous()
    logical_to_physical = _row_map(logical_to_physical, "logical_to_physical", n_tot)
    N.check(N.lib().vsel_topk_select_splice(
        _stream(), hc.data_ptr(), _code(hc), h.shape[1], C.byref(seg), scc.data_ptr(), _p(logical_to_physical),
        ids.data_ptr(), ids.numel(), _p(cu_s), max_len_out, int(visual_token_id), emb.data_ptr(), _p(pos), rows,
        _p(am), o["idx"].data_ptr(), o["sel"].data_ptr(), o["new_ids"].data_ptr(), o["new_emb"].data_ptr(), _p(o["new_pos"]),
        _p(o["new_am"]), o["cu_out"].data_ptr(), o["src"].data_ptr(), o["stats"].data_ptr(), None, None))
    new_am = _select_splice_finish(o, check, n_tot, l_out, k_tot, attention_mask)
    return dict(idx=o["idx"], selected_indices=o["sel"], input_ids=o["new_ids"], inputs_embeds=o["new_emb"],
                position_ids=o["new_pos"], attention_mask=new_am, cu_seqlens=o["cu_out"])


# ------------------------------------------------------------------------------------------------
# var-len attention
# ------------------------------------------------------------------------------------------------

def varlen_attn(q, k, v, cu_seqlens: torch.Tensor, max_seqlen: int, causal: bool = True,
                softmax_scale: Optional[float] = None, key_parts: bool = False) -> torch.Tensor:
    """q [T,Hq,d], k/v [T,Hkv,d] bf16, cu_seqlens int32 [S+1] on device -> out [T,Hq,d].
    key_parts=True: one long prompt (or a few of EQUAL length) on few heads may run as key-range parts through a workspace
    (vsel_varlen_attn_fwd_ws) -- opt-in, because that form sums in another fp32 order than the others (a prompt's result would otherwise
    depend on what it is packed with) and ignores cu_seqlens beyond their count: the equal-length layout is checked here."""
    dev = _dev(q, k, v, cu_seqlens)
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
        raise TypeError("varlen_attn takes bfloat16 q/k/v")
    if cu_seqlens.dtype != torch.int32:
        raise TypeError("cu_seqlens must be int32")
    t, hq, d = q.shape
    hkv = k.shape[1]
    scale = float(softmax_scale) if softmax_scale is not None else d ** -0.5
    out = torch.empty_like(q)
    lib = N.lib()
    n_seq = cu_seqlens.numel() - 1
    # one long prompt (or a few of equal length): key-range parts through a workspace (vsel_varlen_attn_fwd_ws; 0 bytes = not such a batch)
    ws_bytes = lib.vsel_varlen_attn_fwd_workspace_bytes(n_seq, int(max_seqlen), t, hq, hkv, d, int(causal)) if key_parts else 0
    if ws_bytes and n_seq > 1 and not torch.equal(cu_seqlens, torch.arange(0, t + 1, int(max_seqlen), dtype=torch.int32, device=dev)):
        ws_bytes = 0                                  # (total == n_seq * max_seqlen by coincidence: not an equal-length batch)
    if ws_bytes:
        ws = _workspace(ws_bytes, dev)
        N.check(lib.vsel_varlen_attn_fwd_ws(_stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), cu_seqlens.data_ptr(), n_seq, int(max_seqlen),
                                            t, hq, hkv, d, scale, int(causal), out.data_ptr(), None, ws.data_ptr(), ws.numel()))
        return out
    N.check(lib.vsel_varlen_attn_fwd(_stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), cu_seqlens.data_ptr(),
                                     n_seq, int(max_seqlen), t, hq, hkv, d, scale, int(causal),
                                     out.data_ptr()))
    return out


def varlen_attn_fwd_lse(q, k, v, cu_seqlens: torch.Tensor, max_seqlen: int, causal: bool = True,
                        softmax_scale: Optional[float] = None):
    """varlen_attn that also returns lse fp32 [T, Hq] (log-sum-exp of the scaled visible scores) for varlen_attn_bwd."""
    dev = _dev(q, k, v, cu_seqlens)
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
        raise TypeError("varlen_attn takes bfloat16 q/k/v")
    if cu_seqlens.dtype != torch.int32:
        raise TypeError("cu_seqlens must be int32")
    t, hq, d = q.shape
    hkv = k.shape[1]
    scale = float(softmax_scale) if softmax_scale is not None else d ** -0.5
    out = torch.empty_like(q)
    lse = torch.empty(t, hq, dtype=torch.float32, device=dev)
    N.check(N.lib().vsel_varlen_attn_fwd_lse(_stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), cu_seqlens.data_ptr(),
                                             cu_seqlens.numel() - 1, int(max_seqlen), t, hq, hkv, d, scale, int(causal),
                                             out.data_ptr(), lse.data_ptr()))
    return out, lse


def varlen_attn_bwd(dout, q, k, v, out, lse, cu_seqlens: torch.Tensor, max_seqlen: int, causal: bool = True,
                    softmax_scale: Optional[float] = None):
    """Backward of varlen_attn: dout/out/q [T,Hq,d], k/v [T,Hkv,d] bf16 contiguous, lse fp32 [T,Hq] -> (dq, dk, dv) bf16."""
    dev = _dev(dout, q, k, v, out, lse, cu_seqlens)
    for x in (dout, q, k, v, out):
        if x.dtype != torch.bfloat16:
            raise TypeError("varlen_attn_bwd takes bfloat16 tensors")
        if not x.is_contiguous():
            raise ValueError("varlen_attn_bwd takes contiguous tensors")
    if lse.dtype != torch.float32 or cu_seqlens.dtype != torch.int32:
        raise TypeError("lse must be float32 and cu_seqlens int32")
    t, hq, d = q.shape
    hkv = k.shape[1]
    if dout.shape != q.shape or out.shape != q.shape or v.shape != k.shape or lse.shape != (t, hq):
        raise ValueError("varlen_attn_bwd: shape mismatch")
    scale = float(softmax_scale) if softmax_scale is not None else d ** -0.5
    lib = N.lib()
    ws = _workspace(lib.vsel_varlen_attn_bwd_workspace_bytes(t, hq, hkv, cu_seqlens.numel() - 1, int(max_seqlen)), dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    N.check(lib.vsel_varlen_attn_bwd(_stream(), dout.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                                     lse.data_ptr(), cu_seqlens.data_ptr(), cu_seqlens.numel() - 1, int(max_seqlen), t, hq, hkv,
                                     d, scale, int(causal), ws.data_ptr(), ws.numel(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()))
    return dq, dk, dv


def varlen_attn_kv(q, k, v, cu_seqlens_q: torch.Tensor, cu_seqlens_k: torch.Tensor, max_seqlen_q: int, causal: bool = True,
                   softmax_scale: Optional[float] = None) -> torch.Tensor:
    """Separate query / key packings: q [Tq,Hq,d], k/v [Tk,Hkv,d] bf16, cu_seqlens_q / cu_seqlens_k int32 [S+1] ->
    out [Tq,Hq,d].  Bottom-right aligned causal mask (query i of a sequence sees keys <= i + klen - qlen)."""
    dev = _dev(q, k, v, cu_seqlens_q, cu_seqlens_k)
    if q.dtype != torch.bfloat16 or k.dtype != torch.bfloat16 or v.dtype != torch.bfloat16:
        raise TypeError("varlen_attn_kv takes bfloat16 q/k/v")
    if cu_seqlens_q.dtype != torch.int32 or cu_seqlens_k.dtype != torch.int32 or cu_seqlens_q.numel() != cu_seqlens_k.numel():
        raise TypeError("cu_seqlens_q / cu_seqlens_k must be int32 of the same length")
    tq, hq, d = q.shape
    hkv = k.shape[1]
    scale = float(softmax_scale) if softmax_scale is not None else d ** -0.5
    seqlens_k = (cu_seqlens_k[1:] - cu_seqlens_k[:-1]).contiguous()
    out = torch.empty_like(q)
    N.check(N.lib().vsel_varlen_attn_fwd_kv(_stream(), q.data_ptr(), k.data_ptr(), v.data_ptr(), cu_seqlens_q.data_ptr(),
                                            cu_seqlens_k.data_ptr(), seqlens_k.data_ptr(), cu_seqlens_q.numel() - 1,
                                            int(max_seqlen_q), hq, hkv, d, scale, int(causal), out.data_ptr()))
    return out


_head_major_meta: dict = {}


def head_major_ok(t: torch.Tensor) -> bool:
    """[B, H, L, d] view the strided attention entry can address: unit feature stride, 16-byte aligned row / head strides, and
    the batch stride of a dense tensor (the sequence base is cu[b] * H * d)."""
    return (t.dim() == 4 and t.stride(3) == 1 and t.stride(1) % 8 == 0 and t.stride(2) % 8 == 0 and t.stride(1) >= t.shape[3]
            and t.stride(2) >= t.shape[3] and (t.shape[0] == 1 or t.stride(0) == t.shape[1] * t.shape[2] * t.shape[3])
            and t.data_ptr() % 16 == 0)


def attn_head_major(query, key, value, causal: bool = True, softmax_scale: Optional[float] = None) -> torch.Tensor:
    """HuggingFace layout without copies: query [B, Hq, Lq, d], key / value [B, Hkv, Lk, d] bf16, either contiguous in that
    layout or a transposed view of a packed [B, L, H, d] tensor (what HF hands over for v) -> out [B, Lq, Hq, d].  Every batch
    row is one sequence; Lq < Lk: bottom-right aligned causal mask (decode / chunked prefill against the cache)."""
    dev = _dev(query, key, value, strided_ok=True)            # strides are validated by head_major_ok below
    if query.dtype != torch.bfloat16 or key.dtype != torch.bfloat16 or value.dtype != torch.bfloat16:
        raise TypeError("attn_head_major takes bfloat16 tensors")
    b, hq, lq, d = query.shape
    hkv, lk = key.shape[1], key.shape[2]
    if value.shape != key.shape or not all(head_major_ok(t) for t in (query, key, value)):
        raise ValueError("attn_head_major takes [B, H, L, d] tensors with unit feature stride and a dense batch stride")
    scale = float(softmax_scale) if softmax_scale is not None else d ** -0.5
    # the index tensors are the same for every layer of a forward: build them once per (B, Lq, Lk, device)
    key_ = (b, lq, lk, dev)
    if _head_major_meta.get("key") != key_:
        cu_q = torch.arange(0, (b + 1) * lq, lq, dtype=torch.int32, device=dev)
        cu_k = seqlens_k = None
        if lk != lq:
            cu_k = torch.arange(0, (b + 1) * lk, lk, dtype=torch.int32, device=dev)
            seqlens_k = torch.full((b,), lk, dtype=torch.int32, device=dev)
        _head_major_meta["key"], _head_major_meta["val"] = key_, (cu_q, cu_k, seqlens_k)
    cu_q, cu_k, seqlens_k = _head_major_meta["val"]
    out = torch.empty(b, lq, hq, d, dtype=query.dtype, device=dev)
    N.check(N.lib().vsel_varlen_attn_fwd_strided(_stream(), query.data_ptr(), key.data_ptr(), value.data_ptr(), cu_q.data_ptr(),
                                                 _p(cu_k), _p(seqlens_k), b, lq, hq, hkv, d, query.stride(2), query.stride(1),
                                                 key.stride(2), key.stride(1), value.stride(2), value.stride(1), scale,
                                                 int(causal), out.data_ptr()))
    return out


def paged_attn(q, k_cache, v_cache, cu_seqlens_q: torch.Tensor, seqlens_k: torch.Tensor, block_table: torch.Tensor,
               max_seqlen_q: int, causal: bool = True, softmax_scale: Optional[float] = None) -> torch.Tensor:
    """q [Tq,Hq,d] bf16; k_cache / v_cache [n_pages, page_size, Hkv, d] bf16; cu_seqlens_q int32 [S+1];
    seqlens_k int32 [S]; block_table int32 [S, max_pages] -> out [Tq,Hq,d].  Bottom-right aligned causal mask."""
    dev = _dev(q, k_cache, v_cache, cu_seqlens_q, seqlens_k, block_table)
    if q.dtype != torch.bfloat16 or k_cache.dtype != torch.bfloat16 or v_cache.dtype != torch.bfloat16:
        raise TypeError("paged_attn takes bfloat16 q / k_cache / v_cache")
    for t in (cu_seqlens_q, seqlens_k, block_table):
        if t.dtype != torch.int32:
            raise TypeError("cu_seqlens_q / seqlens_k / block_table must be int32")
    tq, hq, d = q.shape
    n_pages, page_size, hkv, _ = k_cache.shape
    n_seq = seqlens_k.numel()
    if block_table.dim() != 2 or block_table.shape[0] != n_seq or cu_seqlens_q.numel() != n_seq + 1:
        raise ValueError("block_table must be [n_seq, max_pages] and cu_seqlens_q [n_seq + 1]")
    scale = float(softmax_scale) if softmax_scale is not None else d ** -0.5
    out = torch.empty_like(q)
    N.check(N.lib().vsel_paged_attn_fwd(_stream(), q.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), cu_seqlens_q.data_ptr(),
                                        seqlens_k.data_ptr(), block_table.data_ptr(), block_table.shape[1], page_size, n_seq,
                                        int(max_seqlen_q), hq, hkv, d, scale, int(causal), out.data_ptr()))
    return out


# every op that reaches libvsel switches to its tensors' device first
for _name in ("lis_scores", "lis_select", "lis_select_permuted", "gelu_colsum", "colsum_linear", "lis_select_presummed", "lis_select_varlen",
              "hard_topk", "gather_rows", "soft_topk_fwd", "soft_topk_bwd", "lis_train_fwd", "lis_train_bwd", "lis_train_bwd_factors", "factors_to_grads", "lis_scores_bwd",
              "splice", "splice_batched", "lis_select_splice", "topk_select_splice", "varlen_attn", "varlen_attn_fwd_lse", "varlen_attn_bwd", "varlen_attn_kv",
              "attn_head_major", "paged_attn"):
    globals()[_name] = _device_guard(globals()[_name])
del _name
