"""Mirror of rec_retrieval/module/recommender/module.py:21-361: forward dispatch, item encoding, full-catalog scoring and the
test / validation loops (merged-model inference), and the fine-tuning side of RecModule -- negative-sampling scores
(:79-131), ``training_step`` (:168-189) and ``configure_optimizers`` (:44-72).  RecJointModule is not built.

RecModule is a LightningModule when ``lightning`` is importable, otherwise a hook-compatible nn.Module
driven by mergerec_amd.utils.Trainer -- the hooks and their order are the same either way."""
from __future__ import annotations

from typing import Dict, List, Literal, Optional

import torch
from torch import nn

from .. import ops
from ..engine import check_module_inputs
from ..autograd import cross_entropy_rows, matmul_nt
from ..configs import NegativeSampleOption
from ..evaluator import Evaluator
from ..model_batch import BatchItem, BatchSequence, BatchSequenceWithNegative

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
            # device scalars stay on the device (no host sync per step); read them with float() when needed
            self._logged[name] = value.detach() if isinstance(value, torch.Tensor) else float(value)

        def log_dict(self, d, **_):
            for k, v in d.items():
                self._logged[k] = float(v)


__all__ = ["RecModule"]


def score_matrix(users: torch.Tensor, items: torch.Tensor, device, rows_per_call: int = 4096) -> torch.Tensor:
    """(U, M) fp32 host matrix ``users @ items.T`` through the exact-fp32 k-ordered GEMM of the scoring path (module.py:137), a block of
    user rows at a time."""
    E = items.to(device, torch.float32).contiguous()
    out = torch.empty(users.shape[0], E.shape[0], dtype=torch.float32)
    for r0 in range(0, users.shape[0], rows_per_call):
        u = users[r0:r0 + rows_per_call].to(device, torch.float32).contiguous()
        out[r0:r0 + u.shape[0]] = ops.gemm_nt(u, [E]).cpu()
    return out


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
        self._eval_scores, self._eval_scores_lazy = None, None
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
        self._shard = None  # parallel.ShardedLoader of the running evaluation epoch (several ranks), set by utils.Trainer.test

    # -- forward (module.py:74-77, 133-166) --------------------------------------------------------
    def _encode(self, batch) -> torch.Tensor:
        if torch.is_grad_enabled() and self.training and hasattr(self.model, "forward_with_grad"):
            # fine-tuning: the exact-fp32 training graph, then F.normalize on the (B, d) rows under autograd
            out = self.model.forward_with_grad(batch)
            return nn.functional.normalize(out, p=2, dim=-1) if self.similarity == "cosine" else out
        if hasattr(self.model, "encode_normalized"):
            return self.model.encode_normalized(batch, normalize=self.similarity == "cosine")
        out = self.model.forward(batch)  # e.g. a TaskVectorMergingModule (re-merges, then encodes)
        if self.similarity == "cosine":
            out = ops.cls_pool_normalize(out.contiguous(), None, out.shape[0], True)
        return out

    def _forward_all_negative(self, batch, labels: torch.Tensor):
        user = self._encode(batch)
        E = self.item_embeddings.data
        scores = matmul_nt(user, E) if user.requires_grad else ops.gemm_nt(user, [E])  # user @ item_embeddings.T (module.py:137)
        return scores, labels, user

    def _forward_negative_sample(self, sequence_batch, target_batch, negative_batch):
        """module.py:79-131.  The reference encodes the three batches one after the other; here they go through ONE packed
        encoder pass (padding never reaches a kernel), and every score the modes need comes from one GEMM ``user @ [target; negatives].T``."""
        from ..data import _cat_encodings

        mode = self.negative_sample.mode
        need_neg = mode in (NegativeSampleOption.SAMPLE, NegativeSampleOption.IN_BATCH_SAMPLE)
        if need_neg:
            assert negative_batch is not None, f"negative_batch must not be None in {mode.name.lower()} mode"
        elif mode != NegativeSampleOption.IN_BATCH:
            raise ValueError(f"Invalid negative sample mode: {mode}")
        parts = [sequence_batch, target_batch] + ([negative_batch] if need_neg else [])
        reps = self._encode(_cat_encodings([dict(p) for p in parts], getattr(self.model.spec, "pad_id", 1)))
        B = sequence_batch["input_ids"].shape[0]
        user, others = reps[:B], reps[B:]
        full = matmul_nt(user, others)  # (B, B [+ B k])
        dev = full.device
        if mode == NegativeSampleOption.IN_BATCH:
            return full, torch.arange(B, device=dev)
        k = self.negative_sample.k
        own_neg = B + torch.arange(B, device=dev).view(B, 1) * k + torch.arange(k, device=dev).view(1, k)  # columns of row b's negatives
        neg_scores = torch.gather(full, 1, own_neg)
        if mode == NegativeSampleOption.SAMPLE:
            pos = torch.gather(full, 1, torch.arange(B, device=dev).view(B, 1))
            return torch.cat((pos, neg_scores), dim=1), torch.zeros(B, dtype=torch.long, device=dev)
        return torch.cat((full[:, :B], neg_scores), dim=1), torch.arange(B, device=dev)

    def _forward_item_encoding(self, batch):
        assert "labels" not in batch, "labels must not be in batch when encoding items"
        return self._encode(batch)

    def forward(self, batch):
        if isinstance(batch, BatchItem):
            return self._forward_item_encoding(batch.items)
        if isinstance(batch, BatchSequence):
            return self._forward_all_negative(batch.sequence, batch.labels)
        if isinstance(batch, BatchSequenceWithNegative):
            return self._forward_negative_sample(batch.sequence, batch.target, batch.negatives)
        raise ValueError(f"Invalid batch type {type(batch)}")

    # -- fine-tuning (module.py:44-72, 168-189) ----------------------------------------------------
    def training_step(self, batch, batch_idx: int):
        output = self.forward(batch)
        if len(output) == 2:
            scores, labels = output
        elif len(output) == 3:
            scores, labels, _ = output
        else:
            raise ValueError(f"Invalid output length {len(output)}")
        loss = cross_entropy_rows(scores / self.temperature, labels)
        self.log("train/loss", loss.detach(), on_step=True, on_epoch=True, prog_bar=True)
        return loss

    def configure_optimizers(self):
        """AdamW with decay on everything but biases / LayerNorm weights + linear warm-up / decay, as ONE fused arena step
        (optim.ArenaAdamW).  ``warmup_steps``: int = steps, float = fraction of trainer.estimated_stepping_batches."""
        from ..optim import ArenaAdamW

        total = self.trainer.estimated_stepping_batches
        if isinstance(self.warmup_steps, float):
            warmup = total * self.warmup_steps
        elif isinstance(self.warmup_steps, int):
            warmup = self.warmup_steps
        else:
            raise ValueError(f"Invalid warmup_steps type {type(self.warmup_steps)}")
        return ArenaAdamW(self.model.train_leaf().detach(), self.model._weights.layout, lr=self.learning_rate, weight_decay=self.weight_decay,
                          num_warmup_steps=warmup, num_training_steps=total,
                          max_grad_norm=getattr(self.trainer, "gradient_clip_val", None))

    # -- the (users, items) score matrix of the last evaluation epoch (module.py:344-352 keeps it on the host, always) -----------------
    @property
    def eval_scores(self):
        """The reference's ``eval_scores``: ALWAYS available after an evaluation epoch.  With ``keep_scores`` the matrix was written by
        the scoring kernel while the epoch ran; otherwise (the fused / staged scoring kept only top-k, log-sum-exp and label ranks) it is
        produced on first access from the epoch's user embeddings and the item matrix by the same exact-fp32 k-ordered product
        (``ops.gemm_nt``) -- the very bits the scoring kernel ranked -- and cached."""
        if self._eval_scores is None and self._eval_scores_lazy is not None:
            users, items = self._eval_scores_lazy
            self._eval_scores = score_matrix(users, items, self.device)
            self._eval_scores_lazy = None
        return self._eval_scores

    @eval_scores.setter
    def eval_scores(self, value):
        self._eval_scores, self._eval_scores_lazy = value, None

    # -- evaluation loops (module.py:283-361; validation = the test loop under the "val/" prefix) ----
    def _eval_start(self):
        self.eval_scores, self.eval_labels, self.eval_user_embeddings = [], [], []
        self._ranks, self._lse, self._lab, self.eval_topk_indices = [], [], [], []
        self._shard = None

    def _eval_step(self, batch: BatchSequence):
        user = self._encode(batch.sequence)
        labels = batch.labels.to(user.device, torch.int64).contiguous()
        k = self.evaluator._max_k
        if k > self.item_embeddings.shape[0]:  # torch.topk(scores, max(ks)) upstream (evaluator/evaluator.py:43) raises the same way
            raise RuntimeError(f"selected index k out of range (max(ks) = {k}, catalog of {self.item_embeddings.shape[0]} items)")
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

    def _eval_end(self, prefix: str, loss_key: str):
        """Epoch end of the test / validation loop.  With several ranks (``self._shard`` set by the Trainer: each rank stepped through
        its own share of the users) the per-user results are put back into the dataloader's row order on every rank first -- two
        all-gathers: one float block (lse, label logit, user embedding), one int block (label rank, label, top-k ids) -- so the
        metrics, ranked indices, label ranks, label logits, embeddings and every ``eval_*`` attribute are those of the single-process run
        bit for bit.  The loss (mean of log-sum-exp minus label logit) agrees to rounding: a shard's batches hold fewer users, and the
        scoring entry picks its route per call from users x M (selection inside the scoring kernel beyond 128 MB of scores, the staged
        pair below), whose log-sum-exp sums differ in order -- at catalog sizes where every batch is staged anyway it is bit-equal too."""
        cat = lambda xs, empty: torch.cat(xs, dim=0) if xs else empty
        dev = self.device
        d = self.item_embeddings.shape[1] if self.item_embeddings is not None else 0
        k = self.evaluator._max_k if self.item_embeddings is not None else 0
        labels = cat(self.eval_labels, torch.empty(0, dtype=torch.int64, device=dev))
        users = cat(self.eval_user_embeddings, torch.empty(0, d, device=dev))
        topk = cat(self.eval_topk_indices, torch.empty(0, k, dtype=torch.int64, device=dev))
        ranks = cat(self._ranks, torch.empty(0, dtype=torch.int32, device=dev))
        lse, lab = cat(self._lse, torch.empty(0, device=dev)), cat(self._lab, torch.empty(0, device=dev))
        scores = torch.cat(self.eval_scores, dim=0) if (self.keep_scores and self.eval_scores) else None
        check_module_inputs(self)  # deferred input checks of every encoder under this module (also a merging module's)
        shard = getattr(self, "_shard", None)
        if shard is not None and shard.world > 1:
            fl = shard.gather_rows(torch.cat([lse.view(-1, 1), lab.view(-1, 1), users], dim=1))
            it = shard.gather_rows(torch.cat([ranks.view(-1, 1).to(torch.int64), labels.view(-1, 1), topk], dim=1))
            lse, lab, users = fl[:, 0].contiguous(), fl[:, 1].contiguous(), fl[:, 2:].contiguous()
            ranks, labels, topk = it[:, 0].to(torch.int32).contiguous(), it[:, 1].contiguous(), it[:, 2:].contiguous()
            if self.keep_scores:  # (U, M) fp32: gathered in column chunks through the device, kept on rank 0 only
                M = self.item_embeddings.shape[0]
                local = scores if scores is not None else torch.empty(0, M)
                parts = [shard.gather_rows(local[:, c0:c0 + 4096].to(dev)) for c0 in range(0, M, 4096)]
                scores = torch.cat([p_.cpu() for p_ in parts], dim=1) if shard.rank == 0 else None
        self.eval_labels, self.eval_user_embeddings, self.eval_topk_indices = labels.cpu(), users.cpu(), topk.cpu()
        if self.item_embeddings is not None and self.eval_labels.numel():
            # F.cross_entropy(scores / T, labels) upstream (module.py:356) raises for a class index outside the catalog; the scoring kernel
            # reads no memory for such a label (its logit is NaN, its rank -1), so the check waits for the epoch's labels on the host
            bad = (self.eval_labels < 0) | (self.eval_labels >= self.item_embeddings.shape[0])
            if bool(bad.any()):
                raise IndexError(f"Target {int(self.eval_labels[bad][0])} is out of bounds.")
        self.eval_scores = scores
        if scores is None and self.item_embeddings is not None and (shard is None or shard.world == 1 or shard.rank == 0):
            # materialised on first access from a SNAPSHOT of the table the kernel ranked (a device copy: tens of MB at HBM rate, once
            # per epoch): the live table may be rewritten in place before anyone asks (a catalog refresh, the next domain's encode), and
            # upstream's eval_scores stays what the epoch computed (module.py:344-352) whatever happens to item_embeddings afterwards
            self._eval_scores_lazy = (self.eval_user_embeddings, self.item_embeddings.data.clone())
        # cross_entropy(scores / T, labels) = mean(logsumexp(row / T) - row[label] / T)   (module.py:318, 356)
        loss = float((lse.double() - lab.double()).mean()) if lse.numel() else float("nan")
        metrics = self.evaluator.from_ranks(ranks, metric_prefix=prefix)
        metrics[loss_key] = loss
        self.log_dict(metrics, prog_bar=True)
        return metrics

    def on_validation_epoch_start(self):
        self._eval_start()

    def validation_step(self, batch: BatchSequence, batch_idx: int, dataloader_idx: int = 0):
        self._eval_step(batch)

    def on_validation_epoch_end(self):
        return self._eval_end("val/", "val/epoch_loss")

    def on_test_epoch_start(self):
        self._eval_start()

    def test_step(self, batch: BatchSequence, batch_idx: int, dataloader_idx: int = 0):
        self._eval_step(batch)

    def on_test_epoch_end(self):
        return self._eval_end("test/", "test/loss")
