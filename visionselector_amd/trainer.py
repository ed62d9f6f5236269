"""LIS-only training loop without HF Trainer / DeepSpeed.

Mirrors what the reference's training entry does around the hot path
(qwen-vl-finetune/qwenvl/train/train_qwen_selector.py):
  * set_model(): only `visual.importance_scorer` is trainable (:127-157)
  * ScheduledWeightTrainer.compute_loss(): regularization_weight = start + (end - start) * min(step / max_steps, 1)
    set on the model before every forward, printed on rank 0 (:66-92)
  * AdamW + cosine schedule + grad-norm clip 1.0 (scripts/sft_7b.sh:17,56-60)
  * data parallel: one process per GPU, gradients of the scorer averaged with ONE bucketed all-reduce
    (visionselector_amd.ddp.LisGradSync; RCCL over xGMI with backend "nccl", gloo in the CPU tests)
  * LIS-only checkpoints with the reference's key names (`visual.importance_scorer.{q_proj,k_proj}.{weight,bias}`).
"""
from __future__ import annotations

import math
import os
from typing import Callable, Dict, Iterable, Optional

import torch
import torch.distributed as dist

from .ddp import LisFactorSync, LisGradSync
from .selector import curriculum_weight, factor_sink

SCORER_KEY = "importance_scorer"


def freeze_all_but_scorer(model: torch.nn.Module) -> list:
    """train_qwen_selector.py:127-157 with tune_compressor=True and everything else frozen."""
    trainable = []
    for n, p in model.named_parameters():
        p.requires_grad = SCORER_KEY in n
        if p.requires_grad:
            trainable.append(p)
    if not trainable:
        raise ValueError("model has no importance_scorer parameters (install_selector first)")
    return trainable


def scorer_state_dict(model: torch.nn.Module, prefix: str = "visual.") -> Dict[str, torch.Tensor]:
    """LIS-only checkpoint with the reference's key layout `visual.importance_scorer.*` (Qwen;
    `model.visual.importance_scorer.*` for LLaVA-OV, llava-ov-15/compression_method/modeling_selector.py:340-343)."""
    out = {}
    for k, v in model.state_dict().items():
        if SCORER_KEY in k:
            out[prefix + k[k.index(SCORER_KEY):]] = v.detach().cpu()
    return out


def load_scorer_state_dict(model: torch.nn.Module, sd: Dict[str, torch.Tensor]) -> None:
    """Load a reference-style checkpoint fragment (any prefix before `importance_scorer.`)."""
    own = {k[k.index(SCORER_KEY):]: k for k in model.state_dict() if SCORER_KEY in k}
    mapped = {}
    for k, v in sd.items():
        if SCORER_KEY in k:
            tail = k[k.index(SCORER_KEY):]
            if tail not in own:
                raise KeyError(f"unexpected scorer key {k}")
            mapped[own[tail]] = v
    missing = set(own.values()) - set(mapped)
    if missing:
        raise KeyError(f"missing scorer keys: {sorted(missing)}")
    model.load_state_dict(mapped, strict=False)


def _scorer_params_in_payload_order(model: torch.nn.Module) -> list:
    """(q_proj.weight, q_proj.bias, k_proj.weight, k_proj.bias) of the model's single importance_scorer."""
    found = {}
    for n, p in model.named_parameters():
        if SCORER_KEY in n:
            tail = n[n.index(SCORER_KEY) + len(SCORER_KEY) + 1:]
            if tail in found:
                raise ValueError(f"more than one importance_scorer in the model ({n})")
            found[tail] = p
    try:
        return [found["q_proj.weight"], found["q_proj.bias"], found["k_proj.weight"], found["k_proj.bias"]]
    except KeyError as e:
        raise ValueError(f"importance_scorer lacks {e}") from None


class LisTrainer:
    """Minimal trainer for the scorer: curriculum-annealed constraint weight, AdamW + cosine, clip, DP gradient mean."""

    def __init__(self, model: torch.nn.Module, max_steps: int, lr: float = 5e-5, weight_decay: float = 0.0,
                 reg_weight_start: float = 0.1, reg_weight_end: float = 2.0, max_grad_norm: float = 1.0,
                 warmup_ratio: float = 0.03, group: Optional[dist.ProcessGroup] = None, log: Callable[[str], None] = print,
                 logging_steps: int = 1, exchange: str = "dense"):
        """exchange="dense": the scorer's fp32 gradients live in one flat bucket, ONE all-reduce per optimizer step
        (51.4 MB at 7B).  exchange="factors": every micro-batch's LIS backward leaves a 57 KB rank-1-factor payload row
        (selector.factor_sink), the step all-gathers the rows and every rank rebuilds the mean gradient itself
        (ddp.LisFactorSync); same gradients to fp32 rounding, and no dense [Hd, D] write per micro-batch."""
        if exchange not in ("dense", "factors"):
            raise ValueError("exchange must be 'dense' or 'factors'")
        self.exchange = exchange
        self.model = model
        self.params = freeze_all_but_scorer(model)
        self.max_steps = max_steps
        self.reg_weight_start, self.reg_weight_end = reg_weight_start, reg_weight_end
        self.max_grad_norm = max_grad_norm
        self.opt = torch.optim.AdamW(self.params, lr=lr, weight_decay=weight_decay)
        # transformers.get_cosine_schedule_with_warmup as HF Trainer configures it (lr_scheduler_type "cosine", warmup_ratio 0.03,
        # scripts/sft_7b.sh:57-58): warm-up steps = ceil(ratio * max_steps), lr starts at 0 and rises linearly, then half a cosine
        warm = int(math.ceil(warmup_ratio * max_steps))
        self.logging_steps = max(1, int(logging_steps))

        def lr_lambda(s: int) -> float:
            if s < warm:
                return s / max(1, warm)
            progress = (s - warm) / max(1, max_steps - warm)
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))

        self.sched = torch.optim.lr_scheduler.LambdaLR(self.opt, lr_lambda)
        # fp32 scorer: gradients live in ONE flat bucket (views), so the data-parallel mean is a single all-reduce, no copies
        if exchange == "factors":
            self.sync = LisFactorSync(_scorer_params_in_payload_order(model), group)
        else:
            self.sync = LisGradSync(self.params, group, bucket_view=all(p.dtype == torch.float32 for p in self.params))
        self.sync.broadcast_parameters(0)
        self.global_step = 0
        self.log = log
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def train_step(self, batches: Iterable[dict]) -> float:
        """One optimizer step over the given micro-batches (gradient accumulation).  Returns the mean loss."""
        w = curriculum_weight(self.global_step, self.max_steps, self.reg_weight_start, self.reg_weight_end)
        self.model.regularization_weight = w                                   # train_qwen_selector.py:82-83
        if self.rank == 0 and self.global_step > 0 and self.global_step % self.logging_steps == 0:
            self.log(f"\n[Step {self.global_step}] Set regularization_weight to: {w:.4f}")    # :86-89
        self.sync.zero_grads()
        batches = list(batches)
        total = 0.0
        if self.exchange == "factors":
            with factor_sink(self.sync):
                for b in batches:
                    out = self.model(**b)
                    (out.loss / len(batches)).backward()
                    total += float(out.loss.detach())
            if any(p.grad is not None for p in self.params):
                raise RuntimeError("exchange='factors': a dense gradient reached the scorer outside the LIS training block "
                                   "(only lis_train_block's backward produces factor payloads); use exchange='dense'")
        else:
            for b in batches:
                out = self.model(**b)
                (out.loss / len(batches)).backward()
                total += float(out.loss.detach())
        self.sync.sync()                                                        # mean over the data-parallel ranks
        torch.nn.utils.clip_grad_norm_(self.params, self.max_grad_norm)
        self.opt.step()
        self.sched.step()
        self.global_step += 1
        return total / len(batches)

    def save(self, path: str) -> None:
        if self.rank == 0:
            os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
            torch.save({"scorer": scorer_state_dict(self.model), "global_step": self.global_step,
                        "optimizer": self.opt.state_dict(), "scheduler": self.sched.state_dict()}, path)

    def resume(self, path: str) -> None:
        ck = torch.load(path, map_location="cpu")
        load_scorer_state_dict(self.model, ck["scorer"])
        self.opt.load_state_dict(ck["optimizer"])
        self.sched.load_state_dict(ck["scheduler"])
        self.global_step = int(ck["global_step"])
