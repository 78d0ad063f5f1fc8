"""Mirror of rec_retrieval/module/recommender/module.py:21-361 restricted to merged-model inference
(forward dispatch, item encoding, full-catalog scoring, test loop); the fine-tuning side
(training_step, negative sampling, optimizers, RecJointModule) is out of scope.

RecModule is a LightningModule when ``lightning`` is importable, otherwise a hook-compatible nn.Module
driven by mergerec_amd.utils.Trainer -- the hooks and their order are the same either way."""
from __future__ import annotations

from typing import Dict, List, Literal, Optional

import torch
from torch import nn

from .. import ops
from ..evaluator import Evaluator
from ..model_batch import BatchItem, BatchSequence

try:  # pragma: no cover - lightning is not installed in the build image
    import lightning as L

    _Base = L.LightningModule
except Exception:  # noqa: BLE001
    L = None

    class _Base(nn.Module):
        """The slice of LightningModule the hot path uses."""

        def __init__(self):
            super().__init__()
            self._logged: Dict[str, float] = {}
            self.trainer = None

        @property
        def device(self):
            m = getattr(self, "model", None)
            return getattr(m, "device", torch.device("cuda", torch.cuda.current_device()))

        def log(self, name, value, **_):
            self._logged[name] = float(value)

        def log_dict(self, d, **_):
            for k, v in d.items():
                self._logged[k] = float(v)


__all__ = ["RecModule"]


class RecModule(_Base):
    def __init__(self, model, evaluator: Evaluator, negative_sample=None, similarity: Literal["dot", "cosine"] = "cosine",
                 temperature: float = 0.05, learning_rate: float = 5e-5, warmup_steps: int = 0, weight_decay: float = 0.0):
        super().__init__()
        if similarity not in ("dot", "cosine"):
            raise ValueError(f"Invalid similarity: {similarity}")
        self.model = model
        self.negative_sample = negative_sample
        self.similarity = similarity
        self.tokenizer = getattr(model, "tokenizer", None)
        self.evaluator = evaluator
        self.temperature = temperature
        self.learning_rate, self.warmup_steps, self.weight_decay = learning_rate, warmup_steps, weight_decay
        self.item_embeddings: Optional[nn.Parameter] = None
        self.eval_scores = []
        self.eval_labels = []
        self.eval_user_embeddings = []
        # The reference keeps every (B, M) score block on the host (module.py:344).  The fused path needs
        # only top-k / lse / label rank; full scores are materialised when a caller asks for them.
        self.keep_scores = False
        self._ranks: List[torch.Tensor] = []
        self._lse: List[torch.Tensor] = []
        self._lab: List[torch.Tensor] = []
        self.eval_topk_indices = []

    # -- forward (module.py:74-77, 133-166) --------------------------------------------------------
    def _encode(self, batch) -> torch.Tensor:
        if hasattr(self.model, "encode_normalized"):
            return self.model.encode_normalized(batch, normalize=self.similarity == "cosine")
        out = self.model.forward(batch)  # e.g. a TaskVectorMergingModule (re-merges, then encodes)
        if self.similarity == "cosine":
            ident = torch.arange(out.shape[0] + 1, dtype=torch.int32, device=out.device)
            out = ops.cls_pool_normalize(out.contiguous(), ident, out.shape[0], True)
        return out

    def _forward_all_negative(self, batch, labels: torch.Tensor):
        user = self._encode(batch)
        E = self.item_embeddings.data
        scores = ops.gemm_nt(user, [E])  # scores = user @ item_embeddings.T (module.py:137)
        return scores, labels, user

    def _forward_item_encoding(self, batch):
        assert "labels" not in batch, "labels must not be in batch when encoding items"
        return self._encode(batch)

    def forward(self, batch):
        if isinstance(batch, BatchItem):
            return self._forward_item_encoding(batch.items)
        if isinstance(batch, BatchSequence):
            return self._forward_all_negative(batch.sequence, batch.labels)
        raise ValueError(f"Invalid batch type {type(batch)}")

    # -- test loop (module.py:325-361) ------------------------------------------------------------
    def on_test_epoch_start(self):
        self.eval_scores, self.eval_labels, self.eval_user_embeddings = [], [], []
        self._ranks, self._lse, self._lab, self.eval_topk_indices = [], [], [], []

    def test_step(self, batch: BatchSequence, batch_idx: int, dataloader_idx: int = 0):
        user = self._encode(batch.sequence)
        labels = batch.labels.to(user.device, torch.int64).contiguous()
        k = min(self.evaluator._max_k, self.item_embeddings.shape[0])
        _, idx, lse, lab, rank, scores = ops.score_topk(user, self.item_embeddings.data, k, labels, 1.0 / self.temperature,
                                                        return_scores=self.keep_scores)
        self._ranks.append(rank)
        self._lse.append(lse)
        self._lab.append(lab)
        self.eval_topk_indices.append(idx)
        self.eval_labels.append(labels)
        self.eval_user_embeddings.append(user)
        if self.keep_scores:
            self.eval_scores.append(scores.cpu())

    def on_test_epoch_end(self):
        cat = lambda xs, empty: torch.cat(xs, dim=0) if xs else empty
        dev = self.device
        self.eval_labels = cat(self.eval_labels, torch.empty(0, dtype=torch.int64, device=dev)).cpu()
        self.eval_user_embeddings = cat(self.eval_user_embeddings, torch.empty(0, 0, device=dev)).cpu()
        self.eval_topk_indices = cat(self.eval_topk_indices, torch.empty(0, 0, dtype=torch.int64, device=dev)).cpu()
        self.eval_scores = torch.cat(self.eval_scores, dim=0) if (self.keep_scores and self.eval_scores) else None
        ranks = cat(self._ranks, torch.empty(0, dtype=torch.int32, device=dev))
        lse, lab = cat(self._lse, torch.empty(0, device=dev)), cat(self._lab, torch.empty(0, device=dev))
        # cross_entropy(scores / T, labels) = mean(logsumexp(row / T) - row[label] / T)   (module.py:356)
        loss = float((lse.double() - lab.double()).mean()) if lse.numel() else float("nan")
        metrics = self.evaluator.from_ranks(ranks, metric_prefix="test/")
        metrics["test/loss"] = loss
        self.log_dict(metrics, prog_bar=True)
        return metrics
