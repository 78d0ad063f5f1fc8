"""Mirror of rec_retrieval/module/recommender/loss_fn.py (class names, constructor arguments, factory): every loss is one
launch of the fused row kernel ``mr_distill_loss_rows_f32`` (csrc/distill.hip), forward value and d loss / d merged logits
together; the batch value is the mean of the per-row values, as in the reference (CE "mean", KL "batchmean", MSE "mean").

``forward(merged_model_logits, single_model_logits)`` takes (N, M) fp32 matrices on the GPU.  The gradient flows to
``merged_model_logits`` only (the teacher logits are constants in the reference too)."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from ..merger.enums import LossType

__all__ = [
    "DistillLossBase", "DistillCELoss", "DistillKDLoss", "DistillMSELoss", "DistillAdaMergingLoss", "DistillAdaMergingKDLoss",
    "MergedPseudoLabelLoss", "MergedPseudoLabelKDLoss", "SinglePseudoLabelLoss", "SinglePseudoLabelKDLoss", "DistillPairwiseLoss",
    "DistillListNetLoss", "distill_loss_factory",
]


class _RowLossFn(torch.autograd.Function):
    """mean over rows of the fused per-row loss; backward is the gradient the same launch produced."""

    @staticmethod
    def forward(ctx, z, t, spec):
        z = z.contiguous()
        want = z.requires_grad or torch.is_grad_enabled() and ctx.needs_input_grad[0]
        loss_row, dz = ops.distill_loss_rows(z, None if t is None else t.contiguous(), want_grad=bool(want), grad_scale=1.0 / z.shape[0], **spec)
        ctx.dz = dz
        return loss_row.mean()

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.dz is None:
            return None, None, None
        return ctx.dz * grad_out, None, None


class DistillLossBase(nn.Module):
    """loss_fn.py:20-34.  Subclasses set ``spec`` = the weights of the kernel's six terms."""

    def __init__(self, *args, **kwargs):
        super().__init__()

    def spec(self) -> dict:
        raise NotImplementedError("Subclasses should implement this method.")

    def forward(self, merged_model_logits: torch.Tensor, single_model_logits: torch.Tensor):
        single = single_model_logits.to(merged_model_logits.device) if single_model_logits is not None else None
        return _RowLossFn.apply(merged_model_logits, single, self.spec())

    def row_losses(self, merged_model_logits: torch.Tensor, single_model_logits: torch.Tensor, want_grad: bool = False, grad_scale: float = 1.0):
        """Per-row values (and gradients) without the batch mean -- what DistillSequenceModule groups by dataset."""
        return ops.distill_loss_rows(merged_model_logits, single_model_logits, want_grad=want_grad, grad_scale=grad_scale, **self.spec())


class DistillCELoss(DistillLossBase):  # loss_fn.py:37-44
    def spec(self):
        return dict(label_src=1, w_ce=1.0)


class DistillKDLoss(DistillLossBase):  # loss_fn.py:47-60
    def __init__(self, temperature: float):
        super().__init__()
        self.temperature = temperature

    def spec(self):
        return dict(w_kd=1.0, temperature=self.temperature)


class DistillAdaMergingLoss(DistillLossBase):  # loss_fn.py:63-69
    def spec(self):
        return dict(w_ent=1.0)


class DistillAdaMergingKDLoss(DistillLossBase):  # loss_fn.py:72-88
    def __init__(self, temperature: float, coefficient: float):
        super().__init__()
        self.temperature, self.coefficient = temperature, coefficient

    def spec(self):
        return dict(w_ent=1.0, w_kd=self.coefficient, temperature=self.temperature)


class MergedPseudoLabelLoss(DistillLossBase):  # loss_fn.py:91-104
    def spec(self):
        return dict(label_src=2, w_ce=1.0)


class MergedPseudoLabelKDLoss(DistillKDLoss):  # loss_fn.py:107-125
    def __init__(self, temperature: float, coefficient: float):
        super().__init__(temperature)
        self.coefficient = coefficient

    def spec(self):
        return dict(label_src=2, w_ce=1.0, w_kd=self.coefficient, temperature=self.temperature)


class SinglePseudoLabelLoss(DistillLossBase):  # loss_fn.py:128-142
    def spec(self):
        return dict(label_src=1, w_ce=1.0)


class SinglePseudoLabelKDLoss(DistillKDLoss):  # loss_fn.py:145-163 (BASELINE config 5: T = 0.05, coefficient = 1000)
    def __init__(self, temperature: float, coefficient: float):
        super().__init__(temperature)
        self.coefficient = coefficient

    def spec(self):
        return dict(label_src=1, w_ce=1.0, w_kd=self.coefficient, temperature=self.temperature)


class DistillMSELoss(DistillLossBase):  # loss_fn.py:166-175
    def spec(self):
        return dict(w_mse=1.0)


class DistillPairwiseLoss(DistillLossBase):  # loss_fn.py:178-199
    def __init__(self, margin: float):
        super().__init__()
        self.margin = margin

    def spec(self):
        return dict(w_pair=1.0, margin=self.margin)


class DistillListNetLoss(DistillLossBase):  # loss_fn.py:202-215 (its eps argument is unused in the reference, too)
    def __init__(self, temperature: float, eps: float = 1e-8):
        super().__init__()
        self.temperature, self.eps = temperature, eps

    def spec(self):
        return dict(w_listnet=1.0, temperature=self.temperature)


def distill_loss_factory(loss_type: LossType, temperature: float | None = None, **kwargs) -> DistillLossBase:
    """loss_fn.py:217-267, same error behaviour."""
    if loss_type is LossType.CE:
        return DistillCELoss(**kwargs)
    if loss_type is LossType.KD:
        if temperature is None:
            raise ValueError("Temperature must be provided for KDLoss.")
        return DistillKDLoss(temperature)
    if loss_type is LossType.MSE:
        return DistillMSELoss(**kwargs)
    if loss_type is LossType.ADAMERGING:
        return DistillAdaMergingLoss(**kwargs)
    if loss_type is LossType.ADAMERGING_KD:
        if temperature is None:
            raise ValueError("Temperature must be provided for AdaMergingKDLoss.")
        if "coefficient" not in kwargs:
            raise ValueError("Coefficient must be provided for AdaMergingKDLoss.")
        return DistillAdaMergingKDLoss(temperature, kwargs["coefficient"])
    if loss_type is LossType.MERGED_PSEUDO_LABEL:
        return MergedPseudoLabelLoss()
    if loss_type is LossType.MERGED_PSEUDO_LABEL_KD:
        if temperature is None:
            raise ValueError("Temperature must be provided for MergedPseudoLabelKDLoss.")
        if "coefficient" not in kwargs:
            raise ValueError("Coefficient must be provided for MergedPseudoLabelKDLoss.")
        return MergedPseudoLabelKDLoss(temperature, kwargs["coefficient"])
    if loss_type is LossType.SINGLE_PSEUDO_LABEL:
        return SinglePseudoLabelLoss()
    if loss_type is LossType.SINGLE_PSEUDO_LABEL_KD:
        if temperature is None:
            raise ValueError("Temperature must be provided for SinglePseudoLabelKDLoss.")
        if "coefficient" not in kwargs:
            raise ValueError("Coefficient must be provided for SinglePseudoLabelKDLoss.")
        return SinglePseudoLabelKDLoss(temperature, kwargs["coefficient"])
    raise ValueError(f"Unknown loss type: {loss_type}")
