"""Synthetic "Amazon-shaped" inputs for the hot path (sizes measured from the reference's
datasets/*.tar.gz where present).  Used by bench.py, smoke() and the tests -- there is no network for
real checkpoints or tokenizer vocabularies, so ids are uniform in [3, vocab) with BOS=0 / EOS=2 / PAD=1.

The tensors have exactly the contract of the reference collators' output (int64, right-padded to the
batch max; collator/recommender/recommender.py:27-32,91-100; utils/recformer_utils.py:45-113)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from .model_batch import BatchItem, BatchSequence

CATALOG_SIZES = {  # M; Arts/Instruments/Office/Scientific are placeholders (datasets absent)
    "Pantry": 4968, "Beauty": 12101, "Sports": 18357, "Toys": 11924,
    "Arts": 22855, "Instruments": 10611, "Office": 27932, "Scientific": 5327,
}
TEST_USERS = {"Pantry": 14178, "Beauty": 22363, "Sports": 35598, "Toys": 19412}


def item_token_lengths(n: int, g: torch.Generator) -> torch.Tensor:
    """tokens per item ~ clip(Normal(36, 9), 12, 110)."""
    return torch.clamp((36 + 9 * torch.randn(n, generator=g)).round(), 12, 110).long()


def sequence_item_counts(n: int, g: torch.Generator) -> torch.Tensor:
    """#input items = min(50, ceil(LogNormal(mu=1.6, sigma=0.8)))."""
    x = torch.exp(1.6 + 0.8 * torch.randn(n, generator=g))
    return torch.clamp(x.ceil(), 1, 50).long()


def blair_item_lengths(n: int, g: torch.Generator) -> torch.Tensor:
    return item_token_lengths(n, g) + 2  # BOS + tokens + EOS


def blair_sequence_lengths(n: int, g: torch.Generator, max_seq_len: int = 512) -> torch.Tensor:
    cnt = sequence_item_counts(n, g)
    out = torch.empty(n, dtype=torch.long)
    for i, c in enumerate(cnt.tolist()):
        toks = int(item_token_lengths(c, g).sum()) + 2 * c + 2  # +2 separator tokens per item, BOS, EOS
        out[i] = min(toks, max_seq_len)
    return out


def _ids_from_lengths(lens: torch.Tensor, vocab: int, g: torch.Generator, pad: int = 1) -> Dict[str, torch.Tensor]:
    B, L = lens.numel(), int(lens.max())
    ids = torch.randint(3, vocab, (B, L), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    ids[:, 0] = 0
    ids[torch.arange(B), lens - 1] = torch.where(lens > 1, torch.full_like(lens, 2), torch.zeros_like(lens))
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, pad))
    return {"input_ids": ids, "attention_mask": mask}


def _recformer_fields(enc: Dict[str, torch.Tensor], lens: torch.Tensor, g: torch.Generator, tokens_per_item: int = 38):
    ids, mask = enc["input_ids"], enc["attention_mask"]
    B, L = ids.shape
    pos = torch.arange(L)[None, :].expand(B, L)
    tt = torch.where(((pos - 1) % tokens_per_item) < 3, torch.ones_like(ids), torch.full_like(ids, 2))  # key / value tokens
    tt[:, 0] = 0
    ip = torch.clamp(1 + (pos - 1) // tokens_per_item, max=50)
    ip[:, 0] = 0
    real = mask.bool()
    tt = torch.where(real, tt, torch.full_like(tt, 3))  # collate pads (recformer_utils.py:97,99)
    ip = torch.where(real, ip, torch.zeros_like(ip))
    gm = torch.zeros_like(ids)
    gm[:, 0] = 1
    enc.update(token_type_ids=tt, item_position_ids=ip, global_attention_mask=gm)
    return enc


def make_batches(lens: torch.Tensor, batch_size: int, vocab: int, g: torch.Generator, kind: str = "roberta",
                 labels: Optional[torch.Tensor] = None) -> List:
    """Split into reference-style batches (each padded to its own max length)."""
    out = []
    for s in range(0, lens.numel(), batch_size):
        l = lens[s : s + batch_size]
        enc = _ids_from_lengths(l, vocab, g)
        if kind == "recformer":
            enc = _recformer_fields(enc, l, g)
        out.append(BatchItem(items=enc) if labels is None else BatchSequence(sequence=enc, labels=labels[s : s + batch_size].clone()))
    return out


@dataclass
class SyntheticDomain:
    name: str
    n_items: int
    item_batches: List[BatchItem]
    sequence_batches: List[BatchSequence]
    labels: torch.Tensor


def make_domain(name: str, n_items: int, n_users: int, batch_size: int, vocab: int, seed: int, kind: str = "roberta",
                max_seq_len: int = 512, item_len_scale: float = 1.0) -> SyntheticDomain:
    g = torch.Generator().manual_seed(seed)
    il = blair_item_lengths(n_items, g)
    sl = blair_sequence_lengths(n_users, g, max_seq_len)
    if item_len_scale != 1.0:
        il = torch.clamp((il.float() * item_len_scale).long(), min=2)
        sl = torch.clamp((sl.float() * item_len_scale).long(), min=2)
    labels = torch.randint(0, n_items, (n_users,), generator=g)
    return SyntheticDomain(name, n_items, make_batches(il, batch_size, vocab, g, kind),
                           make_batches(sl, batch_size, vocab, g, kind, labels=labels), labels)
