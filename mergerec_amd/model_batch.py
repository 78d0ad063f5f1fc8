"""Mirror of rec_retrieval/types/model_batch.py:19-66 (the input dataclasses of the hot path and of the distillation step)."""
from __future__ import annotations

from dataclasses import dataclass, fields, replace
from typing import Any, Mapping

import torch


class ToDeviceMixin:
    def to(self, device):
        def _move(obj):
            if isinstance(obj, torch.Tensor):
                return obj.to(device)
            if hasattr(obj, "to") and not isinstance(obj, (dict, list, tuple)):
                return obj.to(device)  # transformers.BatchEncoding
            if isinstance(obj, Mapping):
                return {k: _move(v) for k, v in obj.items()}
            if isinstance(obj, (list, tuple)):
                return type(obj)(_move(v) for v in obj)
            return obj

        return replace(self, **{f.name: _move(getattr(self, f.name)) for f in fields(self)})


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
