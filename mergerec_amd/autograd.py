"""The two differentiable building blocks of the fine-tuning loss on the HIP path: a score GEMM ``A @ B.T`` with gradients to
both operands, and the row-wise cross entropy with explicit labels.  Both are thin ``torch.autograd.Function`` shells around the
kernels of ops.py; torch itself only carries the graph between them (module/recommender/module.py:79-131, 183 are the torch
expressions these stand for: ``user @ target.T``, ``bmm``, ``F.cross_entropy(scores / T, labels)``)."""
from __future__ import annotations

import torch

from . import ops

__all__ = ["matmul_nt", "cross_entropy_rows"]


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


class _MatmulNT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B):
        A, B = A.contiguous(), B.contiguous()
        ctx.save_for_backward(A, B)
        return ops.gemm_nt(A, [B])  # exact fp32, ascending-k FMA chain (the scoring kernel)

    @staticmethod
    def backward(ctx, dS):
        A, B = ctx.saved_tensors
        n, m = dS.shape
        dA = dB = None
        if ctx.needs_input_grad[0]:  # dA = dS @ B: K = m, zero-padded to the GEMM's k-step
            dS_p = torch.nn.functional.pad(dS, (0, _pad16(m) - m)) if m % 16 else dS.contiguous()
            dA = ops.gemm_nt_train(dS_p, ops.transpose_pad(B))
        if ctx.needs_input_grad[1]:  # dB = dS.T @ A: K = n
            dB = ops.gemm_nt_train(ops.transpose_pad(dS.contiguous()), ops.transpose_pad(A))
        return dA, dB


def matmul_nt(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """(n, d) @ (m, d).T -> (n, m) on the HIP GEMM, differentiable w.r.t. A and B."""
    return _MatmulNT.apply(A, B)


class _CrossEntropyRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, labels):
        z = z.contiguous()
        rows = z.shape[0]
        # the fused row-loss kernel takes its label from a teacher row's arg-max: a one-hot teacher carries explicit labels
        onehot = torch.zeros_like(z)
        onehot.scatter_(1, labels.to(z.device, torch.int64).view(-1, 1), 1.0)
        loss_row, dz = ops.distill_loss_rows(z, onehot, label_src=1, w_ce=1.0, want_grad=ctx.needs_input_grad[0], grad_scale=1.0 / max(rows, 1))
        ctx.dz = dz
        return loss_row.sum() / max(rows, 1)

    @staticmethod
    def backward(ctx, grad_out):
        return (None if ctx.dz is None else ctx.dz * grad_out), None


def cross_entropy_rows(z: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    """mean_r CE(z[r], labels[r]) -- ``F.cross_entropy(z, labels)`` -- value and d / d z from one fused launch."""
    return _CrossEntropyRows.apply(z, labels)
