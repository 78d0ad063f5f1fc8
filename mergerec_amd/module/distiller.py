"""Mirror of rec_retrieval/module/distiller/sequence/module.py:16-100 (DistillSequenceModule) and of the teacher-matrix
set-up of merge_train.py:116-126, on the HIP path.

The reference loops over the batch in Python (one (num_items,) matvec + one loss call per sample).  Here the samples are
grouped by dataset: one fp32 GEMM ``reps_g @ E_ds.T`` per dataset (the ascending-k FMA-chain kernel of the scoring path), the
teacher rows ``S_ds[sequence_id]`` gathered on the device, and ONE fused loss launch per group that returns the per-row
losses and d loss / d logits; the batch loss is the mean over all samples, as in module.py:72.

The whole chain is on the device: loss value, d loss / d logits, d loss / d representations here; representations -> merged
parameters through ``engine_train.EncoderTrainGraph`` (BLaIR / RoBERTa and Recformer / Longformer) and merged parameters -> alpha
through ``mr_merge_bwd_alpha_f32``.  (Row normalisation of the teacher embeddings and the index gathers of the per-sample bookkeeping are
plain torch ops: a few (rows, d) tensors per step.)"""
from __future__ import annotations

import os

from typing import List, Literal, Optional, Sequence

import torch
from torch import nn

from .. import ops
from ..model_batch import BatchDistillationSequence, BatchItem, BatchSequenceWithNegative
from .loss_fn import DistillLossBase

__all__ = ["DistillSequenceModule", "teacher_scores"]


def teacher_scores(sequence_embedding: torch.Tensor, item_embedding: torch.Tensor) -> torch.Tensor:
    """merge_train.py:120-126: S = normalise(seq) @ normalise(item).T, rows normalised by x / ||x|| (no eps, as there)."""
    item = (item_embedding / item_embedding.norm(dim=-1, keepdim=True)).contiguous()
    seq = (sequence_embedding / sequence_embedding.norm(dim=-1, keepdim=True)).contiguous()
    return ops.gemm_nt(seq, [item])


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


class _GroupLossFn(torch.autograd.Function):
    """sum over the group's rows of (row loss / batch size), with d / d reps through logits = reps @ E.T."""

    @staticmethod
    def forward(ctx, reps, E, Et_pad, teacher_rows, loss_fn: DistillLossBase, batch_size: int):
        logits = ops.gemm_nt(reps.contiguous(), [E])
        n, M = logits.shape
        dz_pad = None
        if reps.requires_grad:
            dz_pad = torch.zeros(n, _pad16(M), dtype=torch.float32, device=reps.device)  # K of the backward GEMM: multiple of 16
        rows, _ = ops.distill_loss_rows(logits, teacher_rows, grad_scale=1.0 / batch_size, dz=None if dz_pad is None else dz_pad[:, :M],
                                        **loss_fn.spec())
        ctx.dz_pad, ctx.Et_pad = dz_pad, Et_pad
        return rows.sum() / batch_size

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.dz_pad is None:
            return None, None, None, None, None, None
        d_reps = ops.gemm_nt_train(ctx.dz_pad, ctx.Et_pad)  # (n, Mpad) @ (d, Mpad).T: a handful of rows, k = catalog size -> split-K
        return d_reps * grad_out, None, None, None, None, None


class _BatchedLossFn(torch.autograd.Function):
    """module.py:62-72 for the whole batch at once: z = rep @ E_ds.T per dataset group (ops.skinny_scores: a handful of rows against a whole
    catalog is one stream over E, not an MFMA tile job), ONE fused loss launch over all rows with per-row catalog sizes, and in backward
    d rep = dz @ E_ds per group (ops.skinny_scores_bwd).  Values equal the per-group path's up to summation order."""

    @staticmethod
    def forward(ctx, reps_sorted, module, groups, sid_dev, loss_fn: DistillLossBase, batch_size: int):
        dev = reps_sorted.device
        B = reps_sorted.shape[0]
        m_max = max(module._items[d_i].shape[0] for d_i, _, _ in groups)
        ld = m_max
        z = torch.empty(B, ld, dtype=torch.float32, device=dev)
        t = torch.empty(B, ld, dtype=torch.float32, device=dev)
        row_m = torch.empty(B, dtype=torch.int32)
        reps_c = reps_sorted.contiguous()
        for d_i, off, n in groups:
            M = module._items[d_i].shape[0]
            row_m[off:off + n] = M
            ops.skinny_scores(reps_c[off:off + n], module._items[d_i], out=z[off:off + n, :M])
            ops.gather_rows(module.score_embeddings[d_i], sid_dev[off:off + n], out=t[off:off + n, :M])
        row_m_dev = ops.h2d(row_m, dev)
        need_grad = reps_sorted.requires_grad
        dz = torch.empty(B, ld, dtype=torch.float32, device=dev) if need_grad else None
        rows, _ = ops.distill_loss_rows(z[:, :m_max], t[:, :m_max], grad_scale=1.0 / batch_size, dz=None if dz is None else dz[:, :m_max], row_M=row_m_dev,
                                        **loss_fn.spec())
        ctx.dz, ctx.groups, ctx.module = dz, groups, module
        return rows.sum() / batch_size

    @staticmethod
    def backward(ctx, grad_out):
        if ctx.dz is None:
            return None, None, None, None, None, None
        B = ctx.dz.shape[0]
        d = ctx.module._items[ctx.groups[0][0]].shape[1]
        d_reps = torch.empty(B, d, dtype=torch.float32, device=ctx.dz.device)
        for d_i, off, n in ctx.groups:
            M = ctx.module._items[d_i].shape[0]
            ops.skinny_scores_bwd(ctx.dz[off:off + n, :M], ctx.module._items[d_i], out=d_reps[off:off + n])
        return d_reps * grad_out, None, None, None, None, None


class _LazyLog(dict):
    """name -> last logged value; device tensors are converted to float when READ (one synchronisation, at the reader's time)"""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return float(v) if isinstance(v, torch.Tensor) else v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class DistillSequenceModule(nn.Module):
    def __init__(self, merged_model, score_embeddings: List[torch.Tensor], loss_fn: DistillLossBase, similarity: Literal["dot", "cosine"],
                 learning_rate: float = 5e-5, trainable_args_kwargs: Optional[dict] = None, device=None):
        super().__init__()
        self.merged_model = merged_model
        self.loss_fn = loss_fn
        self.similarity = similarity
        self.learning_rate = learning_rate
        self.trainable_args_kwargs = trainable_args_kwargs or {}
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # teacher matrices live in HBM (the reference keeps up to 18k x 18k fp32 per domain on the host and ships one row
        # per sample); 288 GB makes the gather a device-side row copy
        self.score_embeddings = [s.to(self.device, torch.float32).contiguous() for s in score_embeddings]
        self._items: Optional[List[torch.Tensor]] = None
        self._items_t: Optional[List[torch.Tensor]] = None
        self._valid_metrics: list = []
        self.logged: dict = _LazyLog()

    # -- item embeddings (set by the item-encoding callback in the reference) -------------------------
    @property
    def item_embeddings(self):
        return self._items

    @item_embeddings.setter
    def item_embeddings(self, value: Optional[Sequence[torch.Tensor]]):
        if value is None:
            self._items = self._items_t = None
            return
        self._items = [e.detach().to(self.device, torch.float32).contiguous() for e in value]
        self._items_t = []
        for e in self._items:  # E^T with the catalog dimension zero-padded to a multiple of 16 (backward GEMM's K)
            M, d = e.shape
            et = torch.zeros(d, _pad16(M), dtype=torch.float32, device=self.device)
            et[:, :M] = e.T
            self._items_t.append(et)

    def _maybe_normalize(self, matrix: torch.Tensor):
        if self.similarity == "cosine":
            return nn.functional.normalize(matrix, p=2, dim=-1)
        return matrix

    def forward(self, batch):
        if isinstance(batch, BatchSequenceWithNegative):
            return self._forward_sequence_encoding(batch.sequence)
        elif isinstance(batch, BatchDistillationSequence):
            return self._forward_distill(batch)
        elif isinstance(batch, BatchItem):
            return self._forward_sequence_encoding(batch.items)
        raise ValueError(f"Invalid batch type {type(batch)}")

    def _forward_sequence_encoding(self, sequence_batch):
        return self._maybe_normalize(self.merged_model.forward(sequence_batch))

    def distill_loss(self, representations: torch.Tensor, dataset_indexes: Sequence[int], sequence_ids: Sequence[int]) -> torch.Tensor:
        """module.py:62-72 for already-encoded (and normalised) representations."""
        assert self._items is not None, "item_embeddings must be set (ItemEncodingCallback) before the distillation loss"
        B = representations.shape[0]
        ds = torch.as_tensor(list(dataset_indexes), dtype=torch.int64)
        sid = sequence_ids.detach().to("cpu", torch.int64) if isinstance(sequence_ids, torch.Tensor) else torch.as_tensor(list(sequence_ids), dtype=torch.int64)
        # group the batch by dataset on the host, ship ONE index tensor (row permutation | teacher row ids) to the device
        if ds.numel() != B or sid.numel() != B:
            raise ValueError("dataset_indexes / sequence_ids must hold one entry per representation row")
        if B and (int(ds.min()) < 0 or int(ds.max()) >= len(self._items)):
            raise IndexError(f"dataset index out of range (have {len(self._items)} item matrices)")  # upstream: list index error at module.py:66
        rows_of = torch.tensor([t.shape[0] for t in self.score_embeddings], dtype=torch.int64)
        if B and (int(sid.min()) < 0 or bool((sid >= rows_of[ds]).any())):
            # upstream indexes the teacher matrix with the id (module.py:68) and torch raises; the row-gather kernel would read past the matrix
            raise IndexError("sequence id outside its dataset's teacher-score matrix")
        perm = torch.argsort(ds, stable=True)
        counts = torch.bincount(ds, minlength=len(self._items)).tolist()
        idx_dev = ops.h2d(torch.cat([perm, sid[perm]]).to(torch.int32), self.device)
        perm_dev, sid_dev = idx_dev[:B], idx_dev[B:]
        reps_sorted = representations.index_select(0, perm_dev.long())
        groups = [(d_i, off0, n) for d_i, n, off0 in zip(range(len(counts)), counts, [sum(counts[:i]) for i in range(len(counts))]) if n]
        if os.environ.get("MR_DISTILL_BATCHED", "1") != "0":
            # the whole batch through ONE loss launch: logits of every group written into one (B, M_max) block by the skinny scoring
            # kernel (a stream over each catalog), teacher rows gathered beside them, per-row catalog sizes in a small table
            return _BatchedLossFn.apply(reps_sorted, self, groups, sid_dev, self.loss_fn, B)
        total = None
        for d_i, off, n in groups:
            rows = ops.gather_rows(self.score_embeddings[d_i], sid_dev[off:off + n])
            part = _GroupLossFn.apply(reps_sorted[off:off + n], self._items[d_i], self._items_t[d_i], rows, self.loss_fn, B)
            total = part if total is None else total + part
        return total

    def _forward_distill(self, batch: BatchDistillationSequence):
        reps = self._forward_sequence_encoding(batch.sequence)
        ids = getattr(batch, "host_sequence_ids", None)
        return self.distill_loss(reps, batch.dataset_indexes, batch.sequence_ids if ids is None else ids)

    def log(self, name, value, **kwargs):
        # the tensor, not its value: ``float()`` here would stall the host behind the whole encoder forward in every training step (Lightning's
        # ``self.log`` does not synchronise either); ``logged[name]`` converts when somebody reads it
        self.logged[name] = value.detach() if isinstance(value, torch.Tensor) else float(value)

    def training_step(self, batch: BatchDistillationSequence, batch_idx: int):
        loss = self._forward_distill(batch)
        self.log("train/loss", loss)
        return loss

    def on_validation_epoch_start(self) -> None:
        self._valid_metrics = []

    @torch.no_grad()
    def validation_step(self, batch: BatchDistillationSequence, batch_idx: int, dataloader_idx: int = 0):
        loss = self._forward_distill(batch)
        self.log("val/loss", loss)
        self._valid_metrics.append(loss.item())
        return loss

    def on_validation_epoch_end(self) -> None:
        self.log("val/average_loss_epoch", torch.tensor(self._valid_metrics).mean())
        self._valid_metrics = []

    def configure_optimizers(self):
        return torch.optim.Adam(self.merged_model.trainable_parameters(**self.trainable_args_kwargs), lr=self.learning_rate, weight_decay=0.0)
