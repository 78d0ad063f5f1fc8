"""Optimizer of the fine-tuning path: AdamW over the model's parameter ARENA in one fused HIP launch, the linear warm-up /
linear decay schedule, and global-norm clipping folded into the step.

What it replaces (module/recommender/module.py:44-72 ``configure_optimizers``): ``torch.optim.AdamW`` over two parameter groups --
``weight_decay`` on every parameter whose name contains neither "bias" nor "LayerNorm.weight", 0 on the rest -- stepped by
``transformers.get_linear_schedule_with_warmup``; and Lightning's ``gradient_clip_val`` (clip_grad_norm_ on all parameters).
State: two more arenas (exp_avg, exp_avg_sq) in HBM, 12 B per parameter next to the 4 B of the weights."""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from . import ops
from .engine import ArenaLayout

__all__ = ["ArenaAdamW", "linear_schedule_with_warmup"]

NO_DECAY = ("bias", "LayerNorm.weight")


def linear_schedule_with_warmup(step: int, num_warmup_steps: float, num_training_steps: float) -> float:
    """LR multiplier after ``step`` optimizer steps: 0 -> 1 over the warm-up, then linearly to 0 at num_training_steps."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))


class ArenaAdamW:
    def __init__(self, param: torch.Tensor, layout: ArenaLayout, lr: float, weight_decay: float = 0.0, betas=(0.9, 0.999), eps: float = 1e-8,
                 no_decay: Sequence[str] = NO_DECAY, num_warmup_steps: float = 0, num_training_steps: Optional[float] = None,
                 max_grad_norm: Optional[float] = None):
        if param.dim() != 1 or param.numel() < layout.padded_numel:
            raise ValueError("param must be the flat arena of `layout`")
        self.param, self.layout = param, layout
        self.base_lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), betas, float(eps)
        self.num_warmup_steps, self.num_training_steps = num_warmup_steps, num_training_steps
        self.max_grad_norm = max_grad_norm
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)
        self.step_count = 0
        # one segment per run of adjacent tensors with the same decay (arena offsets are multiples of 64 floats)
        starts, wds = [], []
        for name in layout.shapes:
            # buffers the arena carries (Recformer's embeddings.position_ids) are not parameters: never decayed.  Neither is the RoBERTa
            # pooler: CLS pooling never reads it, so its .grad stays None in the reference and torch's AdamW skips it (decay included)
            frozen = name.endswith("position_ids") or ".pooler." in name
            wd = 0.0 if (frozen or any(nd in name for nd in no_decay)) else self.weight_decay
            if not wds or wds[-1] != wd:
                starts.append(layout.offsets[name])
                wds.append(wd)
        starts[0] = 0
        self.seg_off = torch.tensor(starts + [param.numel()], dtype=torch.int64, device=param.device)
        self.seg_wd = torch.tensor(wds, dtype=torch.float32, device=param.device)

    # -- schedule ------------------------------------------------------------------------------------
    @property
    def lr(self) -> float:
        """The rate the NEXT step uses (LambdaLR semantics: multiplier evaluated at the number of steps taken so far)."""
        if self.num_training_steps is None:
            return self.base_lr
        return self.base_lr * linear_schedule_with_warmup(self.step_count, self.num_warmup_steps, self.num_training_steps)

    # -- step ----------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, grad: torch.Tensor) -> float:
        """Clip (if configured) + AdamW with the scheduled rate; returns that rate."""
        lr = self.lr
        self.step_count += 1
        sumsq = ops.sum_squares(grad) if self.max_grad_norm else None
        ops.adamw_step(self.param, grad, self.exp_avg, self.exp_avg_sq, lr=lr, step=self.step_count, betas=self.betas, eps=self.eps,
                       weight_decay=self.weight_decay, seg_off=self.seg_off, seg_wd=self.seg_wd, grad_sumsq=sumsq,
                       max_grad_norm=float(self.max_grad_norm or 0.0))
        return lr

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "base_lr": self.base_lr}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
