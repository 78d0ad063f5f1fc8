"""Hot-path INPUT side: pre-tokenised domains as the reference collators emit them.

The reference's datamodule (JSON loading, text flattening, HF tokenisation; ~1.9 kLoC of CPU string work that needs
a tokenizer vocabulary unavailable offline) is outside the path.  Its OUTPUT contract is what the kernels consume
(SURVEY 8(b)): int64 (B, L) tensors, right-padded to the batch max -- BLaIR: input_ids (BOS 0 / EOS 2 / PAD 1),
attention_mask; Recformer: + token_type_ids (pad 3), item_position_ids (pad 0), global_attention_mask.

A domain file (`<dir>/tokenized.pt`) is a dict:
    {"items": {key: (M, Li) int64}, "sequences": {key: (U, Ls) int64}, "labels": (U,) int64}
row i of "items" is catalog item id i (callbacks.py:34).  `dump_from_reference.md` in INTEGRATION.md shows the
three lines that write it from the reference's DataModule.  `synthetic:<Name>[:M[:U]]` builds an Amazon-shaped one.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterator, Optional

import torch

from .model_batch import BatchItem, BatchSequence
from . import synthetic


class TokenizedBatches:
    """DataLoader stand-in: yields BatchItem / BatchSequence, each batch trimmed to its own max length
    (the reference tokenises with padding=True per batch, collator/recommender/recommender.py:30,93).
    ``rows``: an explicit ascending row-id list instead of the [lo, hi) range (one rank's share, ``parallel.ShardedLoader``)."""

    def __init__(self, enc: Dict[str, torch.Tensor], batch_size: int, labels: Optional[torch.Tensor] = None,
                 lo: int = 0, hi: Optional[int] = None, rows: Optional[torch.Tensor] = None):
        self.enc, self.bs, self.labels = enc, batch_size, labels
        n = enc["input_ids"].shape[0]
        self.lo, self.hi = lo, n if hi is None else hi
        self.rows = rows

    def __len__(self):
        n = self.hi - self.lo if self.rows is None else self.rows.numel()
        return (n + self.bs - 1) // self.bs

    def __iter__(self) -> Iterator:
        n = self.hi - self.lo if self.rows is None else self.rows.numel()
        for s in range(0, n, self.bs):
            e = min(s + self.bs, n)
            sel = slice(self.lo + s, self.lo + e) if self.rows is None else self.rows[s:e]
            mask = self.enc["attention_mask"][sel]
            L = int(mask.sum(1).max()) if e > s else 0
            enc = {k: v[sel][:, :L].contiguous() for k, v in self.enc.items()}
            yield BatchItem(items=enc) if self.labels is None else BatchSequence(sequence=enc, labels=self.labels[sel])


class TokenizedDomain:
    def __init__(self, name: str, items: Dict[str, torch.Tensor], sequences: Dict[str, torch.Tensor], labels: torch.Tensor):
        self.name, self.items, self.sequences, self.labels = name, items, sequences, labels
        m, u = items["input_ids"].shape[0], sequences["input_ids"].shape[0]
        if labels.shape != (u,):
            raise ValueError("labels must be (U,)")
        if u and (int(labels.min()) < 0 or int(labels.max()) >= m):
            raise ValueError("labels must be item ids in [0, M)")

    @property
    def n_items(self):
        return self.items["input_ids"].shape[0]

    @property
    def n_users(self):
        return self.sequences["input_ids"].shape[0]

    def item_dataloader(self, batch_size, lo=0, hi=None):
        return TokenizedBatches(self.items, batch_size, None, lo, hi)

    def sequence_dataloader(self, batch_size, lo=0, hi=None):
        return TokenizedBatches(self.sequences, batch_size, self.labels, lo, hi)

    def save(self, path):
        torch.save({"items": self.items, "sequences": self.sequences, "labels": self.labels}, path)


def _keep_pad_len(out, encs, L: int):
    """merged encodings whose rows came in batches of different widths carry each row's own padded width (``Encoding.host_pad_len``):
    pooling_method="mean" averages over the padded width of the batch a sequence came in, as upstream does"""
    if any(e["input_ids"].shape[1] != L for e in encs) or any(getattr(e, "host_pad_len", None) is not None for e in encs):
        from .model_batch import Encoding

        out = Encoding(out)
        out.host_pad_len = torch.cat([torch.as_tensor(getattr(e, "host_pad_len", None)).to(torch.int64) if getattr(e, "host_pad_len", None) is not None
                                      else torch.full((e["input_ids"].shape[0],), e["input_ids"].shape[1], dtype=torch.int64) for e in encs])
    return out


def _stack(batches, key_of) -> Dict[str, torch.Tensor]:
    encs = [key_of(b) for b in batches]
    L = max(e["input_ids"].shape[1] for e in encs)
    out = {}
    pads = {"input_ids": 1, "attention_mask": 0, "token_type_ids": 3, "item_position_ids": 0, "global_attention_mask": 0}
    for k in encs[0]:
        rows = []
        for e in encs:
            t = e[k]
            if t.shape[1] < L:
                t = torch.nn.functional.pad(t, (0, L - t.shape[1]), value=pads.get(k, 0))
            rows.append(t)
        out[k] = torch.cat(rows)
    return _keep_pad_len(out, encs, L)


def load_domain(spec: str, kind: str = "roberta", vocab: int = 50265, seed: int = 1234) -> TokenizedDomain:
    """`synthetic:<Name>[:M[:U]]` or a directory / file holding tokenized.pt."""
    if str(spec).startswith("synthetic:"):
        parts = str(spec).split(":")
        name = parts[1]
        m = int(parts[2]) if len(parts) > 2 else synthetic.CATALOG_SIZES.get(name, 5000)
        u = int(parts[3]) if len(parts) > 3 else synthetic.TEST_USERS.get(name, 2 * m)
        dom = synthetic.make_domain(name, m, u, 512, vocab, seed, kind)
        return TokenizedDomain(name, _stack(dom.item_batches, lambda b: b.items), _stack(dom.sequence_batches, lambda b: b.sequence), dom.labels)
    p = Path(spec)
    f = p / "tokenized.pt" if p.is_dir() else p
    if not f.exists():
        raise FileNotFoundError(
            f"{f} not found: the tokeniser side of the reference is not part of this build; dump the collated tensors of the "
            "reference DataModule to <data_path>/tokenized.pt (see INTEGRATION.md) or use synthetic:<Name>"
        )
    d = torch.load(f, map_location="cpu")
    return TokenizedDomain(p.name if p.is_dir() else p.stem, d["items"], d["sequences"], d["labels"])


# ---------------------------------------------------------------------------------------------------------------
_PAD_VALUES = {"input_ids": 1, "attention_mask": 0, "token_type_ids": 3, "item_position_ids": 0, "global_attention_mask": 0}


def _cat_encodings(encs, pad_id: int = 1):
    L = max(e["input_ids"].shape[1] for e in encs)
    out = {}
    for k in encs[0].keys():
        rows = []
        for e in encs:
            t = e[k]
            if t.shape[1] < L:
                t = torch.nn.functional.pad(t, (0, L - t.shape[1]), value=pad_id if k == "input_ids" else _PAD_VALUES.get(k, 0))
            rows.append(t)
        out[k] = torch.cat(rows)
    return _keep_pad_len(out, encs, L)


def coalesce_batches(batches, max_tokens: int = 65536, pad_id: int = 1):
    """Merge consecutive BatchItem / BatchSequence objects into larger ones of up to ~max_tokens attended tokens.
    The kernels work on packed tokens (padding never reaches them), so the per-sequence results are unchanged; only
    the launch granularity changes (a 32-sequence batch fills less than one wave of GEMM workgroups on 256 CUs)."""
    buf, tokens = [], 0

    def flush():
        first = buf[0]
        if isinstance(first, BatchItem):
            return BatchItem(items=_cat_encodings([b.items for b in buf], pad_id))
        return BatchSequence(sequence=_cat_encodings([b.sequence for b in buf], pad_id), labels=torch.cat([b.labels for b in buf]))

    for b in batches:
        enc = b.items if isinstance(b, BatchItem) else b.sequence
        n = int(enc["attention_mask"].sum())
        if buf and tokens + n > max_tokens:
            yield flush()
            buf, tokens = [], 0
        buf.append(b)
        tokens += n
    if buf:
        yield flush()
