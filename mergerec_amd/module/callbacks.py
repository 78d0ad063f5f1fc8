"""Trainer callbacks of the merged-inference and alpha-learning loops.

Boundary (names, hook names, constructor arguments, the alpha jsonl line format) follows rec_retrieval/module/callbacks.py:
ItemEncoderMixin :18-50, ItemEncodingCallback :53-64, ItemEncodingNegativeSampleCallback :67-78, MultiDatasetItemEncodingCallback
:81-109, SaveWeightsCallback :139-174, WeightCheckpointCallback :177-205.  The bodies are this build's: the catalog is encoded in
token-sized packed passes and, when ``torch.distributed`` is initialised, each rank encodes only its share of the catalog rows and
ONE all-gather (RCCL) gives every rank the full item matrix (``parallel.ShardedLoader``)."""
from __future__ import annotations

import re
from pathlib import Path
from typing import Iterable, List, Optional
from uuid import uuid4

import torch
from torch import nn


class ItemEncoderMixin:
    """Catalog encoding shared by the callbacks below: row i of the returned matrix is item id i (the dataloader walks the catalog
    in id order, datamodule/recommender/utils.py:29)."""

    @staticmethod
    @torch.no_grad()
    def encode_items(item_dataloader, pl_module) -> torch.Tensor:
        from .. import parallel
        from ..data import coalesce_batches

        if not hasattr(item_dataloader, "__iter__"):
            raise AssertionError("item_dataloader must be a DataLoader instance.")
        was_training = pl_module.training
        pl_module.eval()
        try:
            shard = parallel.ShardedLoader(item_dataloader, balance=False)  # identity on one rank
            budget = getattr(getattr(pl_module, "trainer", None), "coalesce_tokens", 65536)
            stream = coalesce_batches(shard, budget) if budget else shard
            rows = [pl_module.forward(batch.to(pl_module.device)) for batch in stream]
            if rows:
                local = torch.cat(rows, dim=0)
            else:  # a rank may own no rows of a tiny catalog
                local = torch.empty(0, pl_module.model.spec.hidden, dtype=torch.float32, device=pl_module.device)
            table = shard.gather_rows(local)
            from ..engine import check_module_inputs

            check_module_inputs(pl_module)  # a catalog row with an out-of-range id must not become an embedding silently
            return table
        finally:
            pl_module.train(was_training)

    def inject_item_embeddings(self, item_dataloader, pl_module, requires_grad: bool = False):
        table = self.encode_items(item_dataloader, pl_module)
        pl_module.item_embeddings = nn.Parameter(table, requires_grad=requires_grad)


class _SingleCatalogCallback(ItemEncoderMixin):
    """One catalog, supplied (and swapped per domain by ``utils.test_model``) through the mutable ``item_dataloader``."""

    def __init__(self, item_dataloader=None):
        self.item_dataloader = item_dataloader

    def _encode_if_absent(self, pl_module):
        if pl_module.item_embeddings is not None:
            return
        print("[Test - epoch start] Encoding items as no item embeddings are found.")
        self.inject_item_embeddings(self.item_dataloader, pl_module)

    def on_test_epoch_start(self, trainer, pl_module):
        self._encode_if_absent(pl_module)


class ItemEncodingCallback(_SingleCatalogCallback):
    """Full-catalog training re-encodes the catalog at every training-epoch start (frozen inside the epoch); testing encodes it
    only when the module has none (``test_model`` clears it per domain)."""

    def on_train_epoch_start(self, trainer, pl_module):
        print(f"[Train - epoch {trainer.current_epoch} start] Encoding items.")
        self.inject_item_embeddings(self.item_dataloader, pl_module)


class ItemEncodingNegativeSampleCallback(_SingleCatalogCallback):
    """Sampled-negative training needs the catalog for validation (fresh weights each time) and for the final test."""

    def on_validation_epoch_start(self, trainer, pl_module):
        print(f"[Validation - epoch {trainer.current_epoch} start] Encoding items.")
        self.inject_item_embeddings(self.item_dataloader, pl_module)


class MultiDatasetItemEncodingCallback(ItemEncoderMixin):
    """One catalog per domain for the alpha-learning loop: encoded once with the merged model current at the first training epoch,
    constants of the distillation loss afterwards."""

    def __init__(self, item_dataloaders: List[Iterable]):
        self.item_dataloaders = item_dataloaders

    def inject_item_embeddings(self, item_dataloaders, pl_module, requires_grad: bool = False):
        if pl_module.item_embeddings is not None:
            print("Item embeddings already exist in the model. Skipping encoding.")
            return
        total = len(item_dataloaders)
        tables = []
        for n, loader in enumerate(item_dataloaders, start=1):
            print(f"Encoding {n} / {total} datasets.")
            tables.append(self.encode_items(loader, pl_module))
        pl_module.item_embeddings = tables

    def on_train_epoch_start(self, trainer, pl_module):
        print(f"[Train - epoch {trainer.current_epoch} start] Encoding items.")
        self.inject_item_embeddings(self.item_dataloaders, pl_module)

    def on_test_epoch_start(self, trainer, pl_module):
        if pl_module.item_embeddings is None:
            print("[Test - epoch start] Encoding items as no item embeddings are found.")
            self.inject_item_embeddings(self.item_dataloaders, pl_module)


class SaveWeightsCallback:
    """Appends the current alpha to ``<save_dir>/<version>.jsonl`` every ``log_every_steps`` batches, one ``str(dict)`` per line with
    the keys "epoch", "step", "weights" -- the file ``merge_test.py --weight_file`` reads (``utils.load_alpha_file``).  With
    several ranks only rank 0 writes (alpha is identical on every rank)."""

    def __init__(self, version: Optional[str] = None, save_dir="weights", log_every_steps: int = 5):
        from ..parallel import world

        self.version = version if version is not None else uuid4().hex[:8]
        self.save_dir = Path(save_dir)
        self.save_file = self.save_dir / f"{self.version}.jsonl"
        self.log_every_steps = log_every_steps
        self._file_handler = None
        if world()[0] != 0:
            return
        if not self.save_dir.is_dir():
            print(f"{type(self).__name__}: Creating directory {self.save_dir.absolute()}.")
            self.save_dir.mkdir(parents=True, exist_ok=True)
        self._file_handler = self.save_file.open("w", encoding="utf-8")
        print(f"{type(self).__name__}: Weights will be saved to {self.save_file.absolute()}.")

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx: int) -> None:
        if self._file_handler is None or batch_idx % self.log_every_steps:
            return
        record = dict(epoch=trainer.current_epoch, step=trainer.global_step, weights=pl_module.merged_model.serialize_weights())
        self._file_handler.write(str(record) + "\n")

    def on_train_epoch_end(self, trainer, pl_module):
        if self._file_handler is not None:
            self._file_handler.flush()

    def teardown(self, trainer, pl_module, stage: str):
        handle, self._file_handler = self._file_handler, None
        if handle is not None:
            handle.close()


class WeightCheckpointCallback:
    """Remembers the alpha with the lowest mean over the validation metrics whose names fully match the ``monitor`` regex
    (merge_train.py monitors ``val/loss_epoch/dataloader_idx_\\d+``); ``load_weights`` puts it back after training."""

    def __init__(self, monitor: str = "val/loss"):
        self.monitor = monitor
        self.best_score = float("inf")
        self.best_weights = None

    def on_validation_epoch_end(self, trainer, pl_module):
        pattern = re.compile(self.monitor)
        matched = [float(value) for name, value in trainer.callback_metrics.items() if pattern.fullmatch(name)]
        if not matched:
            raise RuntimeError(f"No metrics found matching the monitor pattern: {self.monitor}")
        mean = sum(matched) / len(matched)
        if mean < self.best_score:
            print(f"New best score: {mean}. Saving weights.")
            self.best_score, self.best_weights = mean, pl_module.merged_model.serialize_weights()

    def load_weights(self, pl_module):
        if self.best_weights is None:
            print("No best weights found. Skipping loading.")
            return
        pl_module.merged_model.load_weights_from_dict(self.best_weights)
        print("Weights loaded from the best checkpoint.")
