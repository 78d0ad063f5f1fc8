"""Mirror of rec_retrieval/evaluator (evaluator.py:6-49, metrics.py:1-88, enums.py:7-13).

``Evaluator.__call__(scores, labels, prefix)`` keeps the reference signature; on GPU tensors the top-k
comes from the HIP kernel (canonical tie order).  ``Evaluator.from_ranks`` is the fused path used by
RecModule: the kernel already reports where each label sits in its row's top-k."""
from __future__ import annotations

from enum import Enum
from typing import Dict, List, Sequence

import torch


def _gain_table(max_k: int) -> List[float]:
    # metrics.py:84 -- 1 / float32(log2(idx + 2)) evaluated in float32, then a Python float division
    return [1 / (torch.log2(torch.tensor(i + 2)).item()) for i in range(max_k)]


class BaseMetric:
    METRIC_NAME = None

    def __init__(self, k: int):
        self.k = k

    @property
    def name(self) -> str:
        return f"{self.METRIC_NAME}@{self.k}"

    def from_ranks(self, ranks: Sequence[int]) -> float:
        raise NotImplementedError

    def __call__(self, y_true: torch.Tensor, y_pred: torch.Tensor) -> float:
        """y_true (N,), y_pred (N, C) ranked predictions (metrics.py:11-22)."""
        rows = y_pred[:, : self.k].tolist()
        ranks = [r.index(t) if t in r else -1 for r, t in zip(rows, y_true.tolist())]
        return self.from_ranks(ranks)


class Recall(BaseMetric):
    METRIC_NAME = "Recall"

    def from_ranks(self, ranks):
        vals = [1.0 if 0 <= r < self.k else 0.0 for r in ranks]
        return sum(vals) / len(vals) if vals else 0.0


class NDCG(BaseMetric):
    METRIC_NAME = "NDCG"

    def from_ranks(self, ranks):
        gains = _gain_table(self.k)
        vals = [gains[r] if 0 <= r < self.k else 0.0 for r in ranks]
        return sum(vals) / len(vals) if vals else 0.0


class MetricType(Enum):
    RECALL = ("RECALL", Recall)
    NDCG = ("NDCG", NDCG)

    def __init__(self, metric_name, metric_cls):
        self.metric_name = metric_name
        self.metric_cls = metric_cls


MAX_K = 1024  # == mr_topk_max_k() (include/mergerec_hip.h); checked against the library in tests/test_cabi_symbols.py


class Evaluator:
    def __init__(self, metrics: List[str], ks: List[int]):
        self.metric_names = metrics
        self.ks = ks
        self._max_k = max(ks)
        # `--ks` is a free flag upstream (torch.topk takes any k, evaluator/evaluator.py:43); the HIP row select holds at most MAX_K
        # candidates per row: reject larger values here, where the flag is parsed into an Evaluator, not at the first batch
        if min(ks) < 1 or self._max_k > MAX_K:
            raise ValueError(f"--ks: every k must be in 1..{MAX_K} (got {sorted(ks)}); the top-k kernel keeps at most {MAX_K} candidates per row")
        self._metrics = [MetricType[m].metric_cls(k) for m in metrics for k in ks]

    def evaluate(self, scores, labels, metric_prefix: str = "") -> Dict[str, float]:
        return self(scores, labels, metric_prefix)

    def __call__(self, scores: torch.Tensor, labels: torch.Tensor, metric_prefix: str = "") -> Dict[str, float]:
        if not scores.is_cuda:
            raise ValueError("Evaluator expects GPU score tensors (top-k runs in the HIP kernel; no CPU fallback)")
        from .. import ops

        if self._max_k > scores.shape[1]:  # torch.topk(scores, self._max_k, dim=1) (evaluator/evaluator.py:43)
            raise RuntimeError(f"selected index k out of range (max(ks) = {self._max_k}, {scores.shape[1]} score columns)")
        _, _, _, _, rank = ops.topk_rows(scores.contiguous(), self._max_k, labels.to(scores.device, torch.int64).contiguous())
        return self.from_ranks(rank, metric_prefix)

    def from_ranks(self, label_rank: torch.Tensor, metric_prefix: str = "") -> Dict[str, float]:
        """label_rank[u] = position of the user's label in its canonical top-max_k list, or -1."""
        ranks = label_rank.tolist()
        return {metric_prefix + m.name: m.from_ranks(ranks) for m in self._metrics}
