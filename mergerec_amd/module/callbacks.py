"""Mirror of rec_retrieval/module/callbacks.py:18-78 (ItemEncoderMixin, ItemEncodingCallback, ItemEncodingNegativeSampleCallback), :81-109
(MultiDatasetItemEncodingCallback) and :139-174 (SaveWeightsCallback)."""
from __future__ import annotations

import re
from pathlib import Path
from uuid import uuid4

import torch
from torch import nn


class ItemEncoderMixin:
    @staticmethod
    @torch.no_grad()
    def encode_items(item_dataloader, pl_module) -> torch.Tensor:
        """callbacks.py:18-38: loop the catalog in id order, forward each batch, concatenate; row == item id."""
        assert hasattr(item_dataloader, "__iter__"), "item_dataloader must be a DataLoader instance."
        train_status = pl_module.training
        pl_module.eval()
        out = []
        tokens = getattr(getattr(pl_module, "trainer", None), "coalesce_tokens", 65536)
        if tokens:
            from ..data import coalesce_batches

            item_dataloader = coalesce_batches(item_dataloader, tokens)
        for batch in item_dataloader:
            out.append(pl_module.forward(batch.to(pl_module.device)))
        pl_module.train(train_status)
        return torch.cat(out, dim=0)

    def inject_item_embeddings(self, item_dataloader, pl_module, requires_grad: bool = False):
        pl_module.item_embeddings = nn.Parameter(self.encode_items(item_dataloader, pl_module), requires_grad=requires_grad)


class ItemEncodingCallback(ItemEncoderMixin):
    def __init__(self, item_dataloader=None):
        self.item_dataloader = item_dataloader

    def on_train_epoch_start(self, trainer, pl_module):
        """callbacks.py:57-59: full-catalog training scores against a catalog frozen at the start of each epoch."""
        print(f"[Train - epoch {trainer.current_epoch} start] Encoding items.")
        self.inject_item_embeddings(self.item_dataloader, pl_module)

    def on_test_epoch_start(self, trainer, pl_module):
        if pl_module.item_embeddings is None:
            print("[Test - epoch start] Encoding items as no item embeddings are found.")
            self.inject_item_embeddings(self.item_dataloader, pl_module)


class ItemEncodingNegativeSampleCallback(ItemEncoderMixin):
    """callbacks.py:67-78: negative-sampling fine-tuning needs the catalog only for validation (re-encoded with the current
    weights at every validation epoch) and for the final test."""

    def __init__(self, item_dataloader=None):
        self.item_dataloader = item_dataloader

    def on_validation_epoch_start(self, trainer, pl_module):
        print(f"[Validation - epoch {trainer.current_epoch} start] Encoding items.")
        self.inject_item_embeddings(self.item_dataloader, pl_module)

    def on_test_epoch_start(self, trainer, pl_module):
        if pl_module.item_embeddings is None:
            print("[Test - epoch start] Encoding items as no item embeddings are found.")
            self.inject_item_embeddings(self.item_dataloader, pl_module)


class MultiDatasetItemEncodingCallback(ItemEncoderMixin):
    """callbacks.py:81-109: one catalog per domain, encoded with the CURRENT merged model at the first training epoch (later
    epochs keep them: 'Item embeddings already exist') and treated as constants by the distillation loss."""

    def __init__(self, item_dataloaders):
        self.item_dataloaders = item_dataloaders

    def inject_item_embeddings(self, item_dataloaders, pl_module, requires_grad: bool = False):
        if pl_module.item_embeddings is not None:
            print("Item embeddings already exist in the model. Skipping encoding.")
            return
        embs = []
        for idx, dl in enumerate(item_dataloaders, start=1):
            print(f"Encoding {idx} / {len(item_dataloaders)} datasets.")
            embs.append(self.encode_items(dl, pl_module))
        pl_module.item_embeddings = embs

    def on_train_epoch_start(self, trainer, pl_module):
        print(f"[Train - epoch {trainer.current_epoch} start] Encoding items.")
        self.inject_item_embeddings(self.item_dataloaders, pl_module)

    def on_test_epoch_start(self, trainer, pl_module):
        if pl_module.item_embeddings is None:
            print("[Test - epoch start] Encoding items as no item embeddings are found.")
            self.inject_item_embeddings(self.item_dataloaders, pl_module)


class SaveWeightsCallback:
    """callbacks.py:139-174: one line per logged step, ``str(dict)`` of {"epoch", "step", "weights"} -- the file format
    ``merge_test.py --weight_file`` reads back (merge_test.py:67-68; use ``mergerec_amd.utils.load_alpha_file``)."""

    def __init__(self, version: str | None = None, save_dir: str | Path = "weights", log_every_steps: int = 5):
        if version is None:
            version = str(uuid4())[:8]
        self.version = version
        self.save_dir = Path(save_dir)
        self.save_file = self.save_dir / f"{version}.jsonl"
        self.log_every_steps = log_every_steps
        if not self.save_dir.exists():
            print(f"{self.__class__.__name__}: Creating directory {self.save_dir.absolute()}.")
            self.save_dir.mkdir(parents=True, exist_ok=True)
        self._file_handler = open(self.save_file, "w", encoding="utf-8")
        print(f"{self.__class__.__name__}: Weights will be saved to {self.save_file.absolute()}.")

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx: int) -> None:
        if batch_idx % self.log_every_steps == 0:
            line = {"epoch": trainer.current_epoch, "step": trainer.global_step, "weights": pl_module.merged_model.serialize_weights()}
            self._file_handler.write(f"{line}\n")

    def on_train_epoch_end(self, trainer, pl_module):
        self._file_handler.flush()

    def teardown(self, trainer, pl_module, stage: str):
        if self._file_handler:
            self._file_handler.close()
            self._file_handler = None


class WeightCheckpointCallback:
    """callbacks.py:177-206: keeps the alpha with the lowest mean of the monitored validation metrics (regex over
    ``trainer.callback_metrics``; merge_train.py monitors ``val/loss_epoch/dataloader_idx_\\d+``) and restores it after training."""

    def __init__(self, monitor: str = "val/loss"):
        self.monitor = monitor
        self.best_score = float("inf")
        self.best_weights = None

    def on_validation_epoch_end(self, trainer, pl_module):
        scores = [float(v) for k, v in trainer.callback_metrics.items() if re.fullmatch(self.monitor, k)]
        if len(scores) == 0:
            raise RuntimeError(f"No metrics found matching the monitor pattern: {self.monitor}")
        current = sum(scores) / len(scores)
        if current < self.best_score:
            print(f"New best score: {current}. Saving weights.")
            self.best_score = current
            self.best_weights = pl_module.merged_model.serialize_weights()

    def load_weights(self, pl_module):
        if self.best_weights is not None:
            pl_module.merged_model.load_weights_from_dict(self.best_weights)
            print("Weights loaded from the best checkpoint.")
        else:
            print("No best weights found. Skipping loading.")
