"""Mirror of rec_retrieval/module/callbacks.py:18-64 (ItemEncoderMixin, ItemEncodingCallback)."""
from __future__ import annotations

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

    def on_test_epoch_start(self, trainer, pl_module):
        if pl_module.item_embeddings is None:
            print("[Test - epoch start] Encoding items as no item embeddings are found.")
            self.inject_item_embeddings(self.item_dataloader, pl_module)
