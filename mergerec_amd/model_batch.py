"""The input dataclasses of the hot path and of the distillation step (boundary: rec_retrieval/types/model_batch.py:19-66 -- same
class names, field names and ``.to(device)``)."""
from __future__ import annotations

from dataclasses import dataclass, fields, replace
from typing import Any, Mapping

import torch


class Encoding(dict):
    """A tokenizer output after ``.to(device)``: a plain dict of tensors plus ``host_lens`` -- the per-row attended-token counts,
    taken while ``attention_mask`` was still in host memory.  The packing kernel needs them on the host (they size every launch);
    carrying them along saves the device -> host round trip per batch that re-deriving them from the moved mask would cost."""

    host_lens = None
    host_pad_len = None  # per-row padded width of the batch each row came in (set when batches of different widths are coalesced): what
    # pooling_method="mean" divides by -- upstream averages over its own batch's padded length


def _relocate(value, device):
    if isinstance(value, torch.Tensor):
        return value.to(device)
    if isinstance(value, Mapping):  # dict / transformers.BatchEncoding
        moved = Encoding((k, _relocate(v, device)) for k, v in value.items())
        lens = getattr(value, "host_lens", None)
        mask = value.get("attention_mask") if lens is None else None
        if isinstance(mask, torch.Tensor) and not mask.is_cuda and mask.dim() == 2:
            lens = mask.ne(0).sum(dim=1)
        moved.host_lens = lens
        moved.host_pad_len = getattr(value, "host_pad_len", None)
        return moved
    if isinstance(value, (list, tuple)):
        return type(value)(_relocate(v, device) for v in value)
    return value.to(device) if hasattr(value, "to") else value


class ToDeviceMixin:
    def to(self, device):
        return replace(self, **{f.name: _relocate(getattr(self, f.name), device) for f in fields(self)})


@dataclass
class BatchItem(ToDeviceMixin):
    items: Any  # BatchEncoding / mapping of int64 (B, L) tensors


@dataclass
class BatchSequence(ToDeviceMixin):
    sequence: Any
    labels: torch.Tensor


@dataclass
class BatchSequenceWithNegative(ToDeviceMixin):  # model_batch.py:47-52
    sequence: Any
    target: Any
    negatives: Any = None


@dataclass
class BatchDistillationSequence(ToDeviceMixin):  # model_batch.py:62-66
    dataset_indexes: list
    sequence_ids: list
    sequence: Any

    host_sequence_ids = None  # ``sequence_ids`` as it was in host memory before ``.to(device)`` (a tensor there): the loss groups the batch by
    # dataset on the host, and reading the ids back from the device would stall the host behind the whole encoder forward

    def to(self, device):
        moved = super().to(device)
        ids = self.sequence_ids
        moved.host_sequence_ids = self.host_sequence_ids if self.host_sequence_ids is not None else (
            ids if isinstance(ids, torch.Tensor) and not ids.is_cuda else None)
        return moved
