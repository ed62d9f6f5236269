"""Data-parallel exchange of the LIS gradients (the only trainable parameters; backbone frozen).

Reference: DeepSpeed ZeRO-2/3 reduce-scatter / all-reduce over NCCL driven by HF Trainer
(qwen-vl-finetune/scripts/sft_7b.sh:9-12,71-74 + zero3.json; set_model freezes all but the scorer,
qwen-vl-finetune/qwenvl/train/train_qwen_selector.py:127-157).  Here: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests), ONE flat fp32 bucket holding
{q_proj.weight, q_proj.bias, k_proj.weight, k_proj.bias}.grad -- 12 848 640 elements = 51.4 MB at 7B -- so the exchange
is a single large all-reduce per optimizer step (xGMI is point-to-point: few, large collectives).
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class LisGradSync:
    """Average the scorer's gradients over the data-parallel group with one bucketed all-reduce."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None,
                 bucket_view: bool = False):
        """bucket_view=True (fp32 parameters): every p.grad is a VIEW into the flat bucket (DDP's gradient_as_bucket_view),
        autograd accumulates into it in place and sync() is one all-reduce with no pack / unpack copies.  Clear gradients
        with zero_grads() (or optimizer.zero_grad(set_to_none=False)), not by setting them to None."""
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters to synchronise")
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self._bucket: Optional[torch.Tensor] = None
        self.bucket_view = bool(bucket_view)
        if self.bucket_view:
            if any(p.dtype != torch.float32 for p in self.params):
                raise TypeError("bucket_view needs float32 parameters (the bucket is fp32)")
            self.attach_views()

    def attach_views(self) -> None:
        """(Re)point every p.grad at its slice of the flat fp32 bucket, keeping what it held."""
        bucket = self._flat(self.params[0].device)
        off = 0
        for p in self.params:
            n = p.numel()
            view = bucket[off:off + n].view_as(p)
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view
            off += n

    def views(self) -> List[torch.Tensor]:
        """The bucket slices in parameter order (for kernels that write gradients directly, e.g. ops.lis_train_bwd(out=...))."""
        bucket = self._flat(self.params[0].device)
        out, off = [], 0
        for p in self.params:
            out.append(bucket[off:off + p.numel()].view_as(p))
            off += p.numel()
        return out

    def zero_grads(self) -> None:
        if self.bucket_view:
            self._flat(self.params[0].device).zero_()
            if any(p.grad is None or p.grad.data_ptr() != v.data_ptr() for p, v in zip(self.params, self.views())):
                self.attach_views()
        else:
            for p in self.params:
                p.grad = None

    def _flat(self, device) -> torch.Tensor:
        if self._bucket is None or self._bucket.device != device:
            self._bucket = torch.empty(self.numel, dtype=torch.float32, device=device)
        return self._bucket

    @torch.no_grad()
    def sync(self) -> None:
        """grad <- mean over ranks (what DDP / ZeRO do before the optimizer step).  Missing grads count as zeros."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return
        bucket = self._flat(self.params[0].device)
        if self.bucket_view:
            if any(p.grad is None or p.grad.data_ptr() != v.data_ptr() for p, v in zip(self.params, self.views())):
                self.attach_views()                    # someone replaced a .grad (e.g. zero_grad(set_to_none=True))
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
            bucket.div_(world)
            return
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                bucket[off:off + n].zero_()
            else:
                bucket[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
        bucket.div_(world)
        off = 0
        for p in self.params:
            n = p.numel()
            g = bucket[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n

    @torch.no_grad()
    def broadcast_parameters(self, src: int = 0) -> None:
        """Make every rank start from rank `src`'s scorer (DDP's initial broadcast)."""
        _broadcast(self.params, self.group, src)


def _broadcast(params, group, src: int = 0) -> None:
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for p in params:
        dist.broadcast(p.data, src=src, group=group)


class LisFactorSync:
    """The same exchange with the scorer's weight gradients as rank-1 FACTORS ("rank-1-factor all-gather").

    Each micro-batch's LIS backward yields dWq = a (x) gx and dWk = dk (x) xsum (ops.lis_train_bwd_factors: one payload row of
    2 (Hd + D) + 2 Hd floats = 57 KB at 7B instead of two dense [Hd, D] gradients = 51.4 MB).  A rank collects the payload rows
    of its micro-batches; sync() all-gathers them (world x micro-batches rows -- on xGMI a 57 KB-per-rank all-gather instead of
    a 51 MB all-reduce) and every rank rebuilds the mean gradient itself:
        dWq = A^T GX / world,  dWk = DK^T XS / world        (two [Hd, R] x [R, D] GEMMs, R = world x micro-batches)
        dbq, dbk = column sums of the gathered bias rows / world
    which is what LisGradSync.sync() leaves in p.grad (sum over a rank's micro-batches, mean over ranks), up to fp32 rounding.
    params = (q_proj.weight, q_proj.bias, k_proj.weight, k_proj.bias) in that order."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group: Optional[dist.ProcessGroup] = None,
                 check_counts: bool = True):
        """check_counts: exchange the per-rank row counts (one 8-byte-per-rank all-gather, a host sync) before the payload
        gather and raise on a mismatch; switch off once the data pipeline guarantees equal micro-batch counts."""
        self.params: List[torch.nn.Parameter] = list(params)
        self.check_counts = bool(check_counts)
        if len(self.params) != 4:
            raise ValueError("LisFactorSync takes (q_proj.weight, q_proj.bias, k_proj.weight, k_proj.bias)")
        wq, bq, wk, bk = self.params
        if wq.dim() != 2 or wq.shape != wk.shape or bq.shape != (wq.shape[0],) or bk.shape != (wq.shape[0],):
            raise ValueError("unexpected scorer parameter shapes")
        self.hd, self.d = int(wq.shape[0]), int(wq.shape[1])
        self.row = 2 * (self.hd + self.d) + 2 * self.hd
        self.group = group
        self._buf: Optional[torch.Tensor] = None       # [capacity, row] payload rows of this step's micro-batches
        self._n = 0
        self._gathered: Optional[torch.Tensor] = None
        self._grads: Optional[List[torch.Tensor]] = None

    @staticmethod
    def _same_device(a: torch.device, b: torch.device) -> bool:
        """'cuda' and 'cuda:<current>' are the same device (torch.device('cuda') != torch.device('cuda:0') as objects)."""
        if a.type != b.type:
            return False
        if a.type != "cuda":
            return True
        cur = torch.cuda.current_device()
        return (cur if a.index is None else a.index) == (cur if b.index is None else b.index)

    def new_row(self, device) -> torch.Tensor:
        """A payload row for the next micro-batch (pass it as ops.lis_train_bwd_factors(out=...)).  The buffer grows (x2)
        only when it is full; rows already handed out this step are kept."""
        device = torch.device(device)
        if self._buf is not None and not self._same_device(self._buf.device, device):
            if self._n:
                raise RuntimeError(f"LisFactorSync: payload rows of one step on two devices ({self._buf.device}, {device})")
            self._buf = None
        if self._buf is None:
            self._buf = torch.empty(4, self.row, dtype=torch.float32, device=device)
        elif self._n == self._buf.shape[0]:
            buf = torch.empty(2 * self._buf.shape[0], self.row, dtype=torch.float32, device=self._buf.device)
            buf[:self._n].copy_(self._buf[:self._n])
            self._buf = buf
        self._n += 1
        return self._buf[self._n - 1]

    @torch.no_grad()
    def broadcast_parameters(self, src: int = 0) -> None:
        _broadcast(self.params, self.group, src)

    def add(self, payload: torch.Tensor) -> None:
        if payload.numel() != self.row or payload.dtype != torch.float32:
            raise ValueError(f"payload must be float32 [{self.row}]")
        self.new_row(payload.device).copy_(payload.reshape(-1))

    def zero_grads(self) -> None:
        self._n = 0
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def sync(self) -> None:
        """p.grad <- mean over ranks of the sum over this step's micro-batches (every rank must have added the same number of rows)."""
        if not self._n:
            raise RuntimeError("LisFactorSync.sync(): no micro-batch payload was added")
        local = self._buf[:self._n]                                      # [M, row], contiguous
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world > 1:
            if self.check_counts:
                # all_gather_into_tensor needs the same row count everywhere: a mismatch would hang or corrupt the gather
                counts = torch.full((world,), -1, dtype=torch.int64, device=local.device)
                dist.all_gather_into_tensor(counts, torch.tensor([self._n], dtype=torch.int64, device=local.device),
                                            group=self.group)
                counts = counts.tolist()
                if any(c != self._n for c in counts):
                    raise RuntimeError(f"LisFactorSync.sync(): ranks added different numbers of micro-batch rows: {counts}")
            if self._gathered is None or self._gathered.shape[0] != world * self._n or self._gathered.device != local.device:
                self._gathered = torch.empty(world * self._n, self.row, dtype=torch.float32, device=local.device)
            dist.all_gather_into_tensor(self._gathered, local, group=self.group)
            gathered = self._gathered
        else:
            gathered = local
        hd, d = self.hd, self.d
        inv = 1.0 / world
        if gathered.is_cuda:
            # one launch writes both dense gradients (sums over the rows in order, fp32), one the two bias gradients
            from . import ops
            if self._grads is None or self._grads[0].device != gathered.device:
                self._grads = [torch.empty(hd, d, dtype=torch.float32, device=gathered.device),
                               torch.empty(hd, dtype=torch.float32, device=gathered.device),
                               torch.empty(hd, d, dtype=torch.float32, device=gathered.device),
                               torch.empty(hd, dtype=torch.float32, device=gathered.device)]
            ops.factors_to_grads(gathered, hd, d, inv, out=self._grads)
            grads = self._grads
        else:                                                            # host tensors (the gloo tests of the exchange logic)
            a, gx, dk, xs, dbq, dbk = torch.split(gathered, [hd, d, hd, d, hd, hd], dim=1)
            grads = [(a.t() @ gx) * inv, dbq.sum(0) * inv, (dk.t() @ xs) * inv, dbk.sum(0) * inv]
        for p, g in zip(self.params, grads):
            p.grad = g if g.dtype == p.dtype else g.to(p.dtype)
        self._n = 0


def shard_units(n_units: int, rank: int, world: int) -> range:
    """Round-robin sharding of independent units (images / samples) over ranks -- the eval harness's accelerate DDP
    split (qwen-evaluation/run_selector.sh:11-18), no data-path collective."""
    return range(rank, n_units, world)
