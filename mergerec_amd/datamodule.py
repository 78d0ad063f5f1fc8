"""Input pipeline of the path: dataset JSON -> item / sequence batches.

Mirrors, without Lightning:
  * ``load_json_files``            rec_retrieval/datamodule/recommender/utils.py:7-37
  * ``RecItemDataset`` / ``RecDataset``   rec_retrieval/datamodule/dataset.py:9-53
  * ``SingleItemCollator`` / ``ItemSequenceCollator``     datamodule/collator/recommender/recommender.py:14-128
  * ``RecDataModule``              datamodule/recommender/datamodule.py:17-151 (text flattening :101-114)
  * ``tokenize_item`` / ``concat_tokenized_items`` / ``pad_tokenized_sequences``   datamodule/utils/recformer_utils.py:12-118
  * ``RecformerSingleItemCollator`` / ``RecformerItemSequenceCollator``    datamodule/collator/recommender/recformer.py:14-97
  * ``RecDataModuleForRecformer``  datamodule/recommender/recformer.py:28-140
  * ``DistillSequenceDataModule`` / ``DistillSequenceCollator`` / ``ChainedDataset`` / ``RecItemAsSequenceDataset`` (merge_train.py's data:
    datamodule/distiller/sequence/datamodule.py:20-184, collator/distiller/collator.py:42-91, dataset.py:20-28,56-88)

File formats (one directory per domain): ``train/val/test.json`` user-id -> item-id list (val / test hold only the new
interaction; the full sequences are train + val (+ test)), ``smap.json`` asin -> item id in id order, ``umap.json``,
``meta_data.json`` asin -> {attribute: text}.  Item ids are catalog rows: the item dataloader walks ``smap`` values in order,
so row r of the item-embedding matrix is item id r (callbacks.py:18-38).

The tokenizer is any *local* ``transformers`` tokenizer directory (``load_tokenizer``; the box is offline).  Batches are the
dataclasses of ``model_batch.py`` holding int64 (B, L) tensors -- exactly what ``EncoderRunner.pack`` consumes.  Parity: the
reference's own collators were run on a synthetic dataset with a local tokenizer (tests/golden/g8_datamodule.pt)."""
from __future__ import annotations

import json
import random
from collections import namedtuple
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch.utils.data import DataLoader, Dataset

from .model_batch import BatchItem, BatchSequence, BatchSequenceWithNegative

__all__ = [
    "load_json_files", "load_tokenizer", "RecItemDataset", "RecDataset", "SingleItemCollator", "ItemSequenceCollator", "RecDataModule",
    "TokenizedItem", "TokenizedSequence", "tokenize_item", "concat_tokenized_items", "pad_tokenized_sequences",
    "RecformerSingleItemCollator", "RecformerItemSequenceCollator", "RecDataModuleForRecformer",
    "RecItemAsSequenceDataset", "ChainedDataset", "split_sequences", "sample_popular", "sample_centroid", "DistillSequenceCollator",
    "DistillSequenceDataModule", "RecformerDistillSequenceCollator", "DistillSequenceDataModuleForRecformer",
]

TokenizedItem = namedtuple("TokenizedItem", ["input_ids", "token_type_ids", "attr_type_ids"])
TokenizedSequence = namedtuple(
    "TokenizedSequence", ["input_ids", "token_type_ids", "attr_type_ids", "item_position_ids", "attention_mask", "global_attention_mask"]
)


def load_tokenizer(path):
    """A tokenizer from a local directory (no hub access on the box)."""
    from transformers import AutoTokenizer

    p = Path(path)
    if not p.is_dir():
        raise FileNotFoundError(f"tokenizer directory {p} not found: the box is offline, pass a local tokenizer directory")
    return AutoTokenizer.from_pretrained(str(p), local_files_only=True)


# ------------------------------------------------------------------------------------------------ files and datasets
def _read(path: Path):
    with open(path, "r") as f:
        return json.load(f)


def load_json_files(dataset_path: Path, max_items: int):
    """utils.py:7-37.  Returns (item_dataset, train, val, test, metadata by item id, umap, smap)."""
    dataset_path = Path(dataset_path)
    train_seq, val_seq, test_seq = ({int(k): v for k, v in _read(dataset_path / f"{s}.json").items()} for s in ("train", "val", "test"))
    metadata, umap, smap = _read(dataset_path / "meta_data.json"), _read(dataset_path / "umap.json"), _read(dataset_path / "smap.json")
    # the evaluation sequences are cumulative: val = train + val, test = train + val + test (utils.py:23-26)
    for k in val_seq:
        val_seq[k] = train_seq.get(k, []) + val_seq[k]
    for k in test_seq:
        test_seq[k] = val_seq.get(k, []) + test_seq[k]
    by_id = {smap[asin]: meta for asin, meta in metadata.items() if asin in smap}
    return (RecItemDataset(list(smap.values())), RecDataset(train_seq, False, max_items), RecDataset(val_seq, False, max_items),
            RecDataset(test_seq, False, max_items), by_id, umap, smap)


class RecItemDataset(Dataset):
    """dataset.py:9-17."""

    def __init__(self, items: List[int]):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, index) -> int:
        return self.items[index]


class RecDataset(Dataset):
    """dataset.py:31-53: (index, last max_items + 1 interactions); ``sample`` draws a random prefix (training only)."""

    def __init__(self, sequence: Dict[int, List[int]], sample: bool, max_items: int):
        self.sequence = list(sequence.values())
        self.sample = sample
        self.max_items = max_items

    def __len__(self):
        return len(self.sequence)

    def balance_weights(self):
        """Length proxy per sample for ``parallel.ShardedLoader``'s token-balanced deal (items per sequence), from the stored lists --
        no sample is built; None while prefixes are drawn at random (training)."""
        if self.sample:
            return None
        return [min(len(s), self.max_items + 1) for s in self.sequence]

    def __getitem__(self, index) -> Tuple[int, List[int]]:
        seq = self.sequence[index]
        if self.sample:
            if len(seq) < 2:
                return seq  # (as the reference: a bare list for degenerate sequences)
            seq = seq[: random.randint(2, len(seq))]
        return index, seq[-(self.max_items + 1):]


def _split_sequences(batch, reverse: bool):
    """recommender.py:62-74 / recformer.py:47-59: inputs = all but the last interaction (most recent first when reversed)."""
    inputs, targets = [], []
    for _, seq in batch:
        head = seq[:-1]
        inputs.append(head[::-1] if reverse else head)
        targets.append(seq[-1])
    return inputs, targets


def _encode_texts(owner, texts):
    """``owner.tokenizer(texts, padding=True, truncation=True, return_tensors="pt", max_length=owner.max_seq_len)`` -- through a private
    copy of the fast tokenizer's Rust backend with right-truncation to max_seq_len and pad-to-longest switched on (what that call
    configures per call).  Going to the backend directly skips transformers' python-side BatchEncoding assembly, which costs 4x the
    tokenisation itself (3.5 s of the 5.2 s evaluation loop of a 14 k-user domain was ``convert_to_tensors``): with it the loop is bound by
    the GPU, not by the collator.  Tokenizers without a fast backend / with left padding take the transformers route unchanged."""
    if not hasattr(owner, "_backend"):
        owner._backend = None
        tok = owner.tokenizer
        be = getattr(tok, "backend_tokenizer", None)
        pad_id, pad_tok = getattr(tok, "pad_token_id", None), getattr(tok, "pad_token", None)
        side_ok = getattr(tok, "padding_side", "right") == "right" and getattr(tok, "truncation_side", "right") == "right"
        names = list(getattr(tok, "model_input_names", ["input_ids", "attention_mask"]))
        plain = set(names) <= {"input_ids", "attention_mask"}  # (BERT-style tokenizers also emit token_type_ids: transformers route)
        if be is not None and pad_id is not None and pad_tok is not None and side_ok and plain:
            try:
                from tokenizers import Tokenizer

                fast = Tokenizer.from_str(be.to_str())
                fast.enable_truncation(max_length=owner.max_seq_len)
                fast.enable_padding(pad_id=pad_id, pad_token=pad_tok)
                owner._backend = fast
            except Exception:  # noqa: BLE001 - any tokenizer the shortcut does not understand takes the transformers route
                owner._backend = None
    if owner._backend is None:
        return owner.tokenizer(texts, padding=True, truncation=True, return_tensors="pt", max_length=owner.max_seq_len)
    from transformers import BatchEncoding

    encs = owner._backend.encode_batch(list(texts))
    return BatchEncoding({"input_ids": torch.tensor([e.ids for e in encs], dtype=torch.int64),
                          "attention_mask": torch.tensor([e.attention_mask for e in encs], dtype=torch.int64)})


# ------------------------------------------------------------------------------------------------ text (BLaIR) collators
class SingleItemCollator:
    """recommender.py:14-35."""

    def __init__(self, tokenizer, item_text: Dict[int, str], max_seq_len: int, item_prompt: str = ""):
        self.tokenizer, self.item_text, self.max_seq_len, self.item_prompt = tokenizer, item_text, max_seq_len, item_prompt

    def _encode(self, texts):
        return _encode_texts(self, texts)

    def __call__(self, batch: List[int]) -> BatchItem:
        return BatchItem(items=self._encode([self.item_prompt + self.item_text[i] for i in batch]))


class ItemSequenceCollator(SingleItemCollator):
    """recommender.py:38-128."""

    def __init__(self, tokenizer, item_text: Dict[int, str], max_seq_len: int, num_negative: Optional[int], in_batch_negative: bool,
                 separator: str = "; ", sequence_prompt: str = "", item_prompt: str = "", reverse_sequence: bool = True):
        super().__init__(tokenizer, item_text, max_seq_len, item_prompt)
        self.num_negative, self.in_batch_negative = num_negative, in_batch_negative
        self.separator, self.sequence_prompt, self.reverse_sequence = separator, sequence_prompt, reverse_sequence
        self._all_items = set(item_text.keys())

    def _text(self, items: Sequence[int]) -> str:
        return self.sequence_prompt + self.separator.join(self.item_text[i] for i in items)

    def __call__(self, batch):
        inputs, targets = _split_sequences(batch, self.reverse_sequence)
        if not self.reverse_sequence:
            # oldest-first sequences are cut from the front until they fit (recommender.py:79-92); reversed ones rely on the
            # tokenizer's right truncation, which drops the oldest items
            cut = []
            for items in inputs:
                items = list(items)
                while len(self.tokenizer.tokenize(self._text(items))) > self.max_seq_len:
                    items.pop(0)
                cut.append(items)
            inputs = cut
        sequence = self._encode([self._text(items) for items in inputs])
        if self.num_negative is None and not self.in_batch_negative:
            return BatchSequence(sequence=sequence, labels=torch.tensor(targets, dtype=torch.long))
        target = self._encode([self.item_prompt + self.item_text[t] for t in targets])
        negatives = None
        if self.num_negative is not None:
            neg = []
            for _, items in batch:
                # negatives = catalog items the user never interacted with.  (The reference takes set() of the whole (index, items)
                # sample -- recommender.py:107-109 -- which raises TypeError on the inner list, so its sampled modes cannot run; the
                # evident intent is followed here.)
                neg.extend(random.sample(sorted(self._all_items - set(items)), self.num_negative))
            negatives = self._encode([self.item_prompt + self.item_text[n] for n in neg])
        return BatchSequenceWithNegative(sequence=sequence, target=target, negatives=negatives)


class _DataModuleBase:
    """The dataloader surface shared by both datamodules (datamodule.py:116-151, recformer.py:106-140)."""

    batch_size: int
    num_workers: int

    def _loader(self, dataset, collate, shuffle=False, drop_last=False):
        return DataLoader(dataset, batch_size=self.batch_size, collate_fn=collate, shuffle=shuffle, num_workers=self.num_workers,
                          drop_last=drop_last)

    def item_dataloader(self):
        return self._loader(self.item_dataset, self.item_collator)

    def train_dataloader(self):
        return self._loader(self.train_dataset, self.sequence_train_collator, shuffle=True, drop_last=True)

    def val_dataloader(self):
        return self._loader(self.val_dataset, self.sequence_eval_collator)

    def test_dataloader(self):
        return self._loader(self.test_dataset, self.sequence_eval_collator)


class RecDataModule(_DataModuleBase):
    """datamodule.py:17-151.  ``negative_sample`` needs ``.k`` and ``.in_batch`` (NegativeSampleConfig); None = full catalog."""

    def __init__(self, dataset_path, tokenizer, batch_size: int, max_seq_len: int, max_attribute_len: int, max_items: int,
                 negative_sample=None, num_workers: int = 0, sequence_prompt: Optional[str] = None, item_prompt: Optional[str] = None,
                 reverse_sequence: bool = True):
        self.dataset_path, self.tokenizer = Path(dataset_path), tokenizer
        self.batch_size, self.max_seq_len, self.max_attribute_len, self.max_items = batch_size, max_seq_len, max_attribute_len, max_items
        self.negative_sample, self.num_workers, self.reverse_sequence = negative_sample, num_workers, reverse_sequence
        self.sequence_prompt = sequence_prompt or ""
        self.item_prompt = item_prompt or ""
        self.item_dataset = self.train_dataset = self.val_dataset = self.test_dataset = None
        self.metadata = self.item_text = None
        self.item_collator = self.sequence_train_collator = self.sequence_eval_collator = None

    def setup(self, stage: str = "test"):
        self.item_dataset, self.train_dataset, self.val_dataset, self.test_dataset, self.metadata, _, _ = load_json_files(
            self.dataset_path, self.max_items)
        self.item_text = {item_id: self._flatten_key_value(meta) for item_id, meta in self.metadata.items()}
        self.item_collator = SingleItemCollator(self.tokenizer, self.item_text, self.max_seq_len, self.item_prompt)
        k = getattr(self.negative_sample, "k", None)
        in_batch = getattr(self.negative_sample, "in_batch", False)
        common = dict(sequence_prompt=self.sequence_prompt, item_prompt=self.item_prompt, reverse_sequence=self.reverse_sequence)
        self.sequence_train_collator = ItemSequenceCollator(self.tokenizer, self.item_text, self.max_seq_len, k, in_batch, **common)
        self.sequence_eval_collator = ItemSequenceCollator(self.tokenizer, self.item_text, self.max_seq_len, None, False, **common)

    def _flatten_key_value(self, item_metadata: Dict[str, str]) -> str:
        """datamodule.py:101-114: "key: value" per attribute, the value cut to max_attribute_len TOKENS and detokenised."""
        parts = []
        for key, value in item_metadata.items():
            assert isinstance(value, str), "Item metadata value must be a string"
            tokens = self.tokenizer.tokenize(value)[: self.max_attribute_len]
            parts.append(f"{key}: {self.tokenizer.convert_tokens_to_string(tokens)}")
        return " ".join(parts)


# ------------------------------------------------------------------------------------------------ Recformer (pre-tokenised items)
def tokenize_item(item_metadata: Dict[str, str], tokenizer, attr_name_id_map, max_attribute_len: int) -> TokenizedItem:
    """recformer_utils.py:12-42: per attribute, key tokens (type 1) + value tokens cut to max_attribute_len (type 2); the
    attribute id comes from ``attr_name_id_map`` (first-seen order, starting at 1)."""
    ids, types, attrs = [], [], []
    for key, value in item_metadata.items():
        assert isinstance(value, str), "Item metadata value must be a string"
        k_ids = tokenizer.convert_tokens_to_ids(tokenizer.tokenize(key))
        v_ids = tokenizer.convert_tokens_to_ids(tokenizer.tokenize(value)[:max_attribute_len])
        ids += k_ids + v_ids
        types += [1] * len(k_ids) + [2] * len(v_ids)
        attrs += [attr_name_id_map[key]] * (len(k_ids) + len(v_ids))
    return TokenizedItem(input_ids=ids, token_type_ids=types, attr_type_ids=attrs)


def concat_tokenized_items(tokenized_items: List[TokenizedItem], bos_token_id: int) -> TokenizedSequence:
    """recformer_utils.py:45-69: <s> (type 0, position 0, global attention) then the items, item position = 1, 2, ..."""
    ids, types, attrs, pos = [bos_token_id], [0], [0], [0]
    for p, item in enumerate(tokenized_items, start=1):
        ids += item.input_ids
        types += item.token_type_ids
        attrs += item.attr_type_ids
        pos += [p] * len(item.input_ids)
    n = len(ids)
    return TokenizedSequence(ids, types, attrs, pos, [1] * n, [1] + [0] * (n - 1))


def pad_tokenized_sequences(tokenized_sequences: List[TokenizedSequence], pad_token_id: int, max_length: int,
                            pad_to_multiple_of: Optional[int] = None):
    """recformer_utils.py:72-118: right-truncate to min(longest, max_length), pad with (pad id, type 3, 0, 0, 0, 0)."""
    from transformers import BatchEncoding

    width = min(max(len(s.input_ids) for s in tokenized_sequences), max_length)
    if pad_to_multiple_of and pad_to_multiple_of > 0 and width % pad_to_multiple_of:
        width = min(-(-width // pad_to_multiple_of) * pad_to_multiple_of, max_length)
    fills = dict(input_ids=pad_token_id, token_type_ids=3, attr_type_ids=0, item_position_ids=0, attention_mask=0, global_attention_mask=0)
    data = {}
    for name, fill in fills.items():
        rows = []
        for seq in tokenized_sequences:
            v = getattr(seq, name)[:width]
            rows.append(v + [fill] * (width - len(v)))
        data[name] = torch.tensor(rows, dtype=torch.int64)
    # (tensors built here: BatchEncoding(..., tensor_type="pt") walks every nested list in python -- 10 s of an 11 s evaluation loop
    # on a 14 k-user domain)
    return BatchEncoding(data=data)


class RecformerSingleItemCollator:
    """recformer.py:14-33 (collator)."""

    def __init__(self, bos_token_id: int, pad_token_id: int, tokenized_items: Dict[int, TokenizedItem], max_seq_len: int):
        self.bos_token_id, self.pad_token_id = bos_token_id, pad_token_id
        self.tokenized_items, self.max_seq_len = tokenized_items, max_seq_len

    def _encode(self, groups: List[List[int]]):
        seqs = [concat_tokenized_items([self.tokenized_items[i] for i in g], self.bos_token_id) for g in groups]
        return pad_tokenized_sequences(seqs, self.pad_token_id, self.max_seq_len)

    def __call__(self, batch: List[int]) -> BatchItem:
        return BatchItem(items=self._encode([[i] for i in batch]))


class RecformerItemSequenceCollator(RecformerSingleItemCollator):
    """recformer.py:36-97 (collator): always most-recent-first."""

    def __init__(self, bos_token_id: int, pad_token_id: int, tokenized_items, max_seq_len: int, num_negative: Optional[int],
                 in_batch_negative: bool):
        super().__init__(bos_token_id, pad_token_id, tokenized_items, max_seq_len)
        self.num_negative, self.in_batch_negative = num_negative, in_batch_negative
        self._all_items = set(tokenized_items.keys())

    def __call__(self, batch):
        inputs, targets = _split_sequences(batch, True)
        sequence = self._encode(inputs)
        if self.num_negative is None and not self.in_batch_negative:
            return BatchSequence(sequence=sequence, labels=torch.tensor(targets, dtype=torch.long))
        target = self._encode([[t] for t in targets])
        negatives = None
        if self.num_negative is not None:
            neg = []
            for _, items in batch:  # (same repair as the text collator: the reference's set() of the sample tuple cannot run)
                neg.extend(random.sample(sorted(self._all_items - set(items)), self.num_negative))
            negatives = self._encode([[n] for n in neg])
        return BatchSequenceWithNegative(sequence=sequence, target=target, negatives=negatives)


class _FirstSeenIds(dict):
    """attribute name -> 1, 2, 3, ... in first-seen order (recformer.py:19-25's counter behind a defaultdict)."""

    def __missing__(self, key):
        self[key] = len(self) + 1
        return self[key]


class RecDataModuleForRecformer(_DataModuleBase):
    """recformer.py:28-140 (datamodule)."""

    def __init__(self, dataset_path, tokenizer, batch_size: int, max_seq_len: int, max_attribute_len: int, max_items: int,
                 negative_sample=None, num_workers: int = 0):
        self.dataset_path, self.tokenizer = Path(dataset_path), tokenizer
        self.batch_size, self.max_seq_len, self.max_attribute_len, self.max_items = batch_size, max_seq_len, max_attribute_len, max_items
        self.negative_sample, self.num_workers = negative_sample, num_workers
        self.bos_token_id, self.pad_token_id = tokenizer.bos_token_id, tokenizer.pad_token_id
        self._attr_name_id_map = _FirstSeenIds()
        self.item_dataset = self.train_dataset = self.val_dataset = self.test_dataset = None
        self.metadata = self.tokenized_items = None
        self.item_collator = self.sequence_train_collator = self.sequence_eval_collator = None

    def setup(self, stage: str = "test"):
        self.item_dataset, self.train_dataset, self.val_dataset, self.test_dataset, self.metadata, _, _ = load_json_files(
            self.dataset_path, self.max_items)
        self.tokenized_items = {
            item_id: tokenize_item(meta, self.tokenizer, self._attr_name_id_map, self.max_attribute_len) for item_id, meta in self.metadata.items()
        }
        args = (self.bos_token_id, self.pad_token_id, self.tokenized_items, self.max_seq_len)
        self.item_collator = RecformerSingleItemCollator(*args)
        self.sequence_train_collator = RecformerItemSequenceCollator(*args, getattr(self.negative_sample, "k", None),
                                                                    getattr(self.negative_sample, "in_batch", False))
        self.sequence_eval_collator = RecformerItemSequenceCollator(*args, None, False)


# ------------------------------------------------------------------------------------------------ collaborative-merging data (merge_train.py)
class RecItemAsSequenceDataset(Dataset):
    """dataset.py:20-28: every catalog item as a one-item pseudo-user sequence ``(index, [item, -1])`` (train_data_split "item")."""

    def __init__(self, items: List[int]):
        self.items = items

    def __len__(self):
        return len(self.items)

    def __getitem__(self, index):
        return index, [self.items[index], -1]


class ChainedDataset(Dataset):
    """dataset.py:56-88: concatenation that also yields which dataset a sample came from: (dataset index, sample)."""

    def __init__(self, datasets: list, start_dataset_idx: int = 0):
        self.datasets = datasets
        self.cumulative_sizes = []
        total = 0
        for d in datasets:
            total += len(d)
            self.cumulative_sizes.append(total)
        self.start_dataset_idx = start_dataset_idx

    def __len__(self):
        return self.cumulative_sizes[-1]

    def __getitem__(self, idx):
        if idx < 0:
            if -idx > len(self):
                raise ValueError("Index out of range")
            idx += len(self)
        from bisect import bisect_right

        k = bisect_right(self.cumulative_sizes, idx)
        local = idx if k == 0 else idx - self.cumulative_sizes[k - 1]
        return k + self.start_dataset_idx, self.datasets[k][local]


def split_sequences(score_dataset, valid_ratio: Optional[float] = None):
    """distiller/sequence/utils.py:32-46 (one torch.randperm draw when a validation share is asked for)."""
    from torch.utils.data import Subset

    if valid_ratio is None:
        return score_dataset, None
    perm = torch.randperm(len(score_dataset))
    cut = int(len(score_dataset) * (1 - valid_ratio))
    return Subset(score_dataset, perm[:cut]), Subset(score_dataset, perm[cut:])


def sample_popular(test_sequence, num_sequences: int):
    """distiller/sequence/utils.py:14-29: the most frequent items of the test sequences."""
    from collections import Counter

    counter = Counter()
    for seq in test_sequence:
        counter.update(seq)
    return [item for item, _ in counter.most_common(num_sequences)]


def sample_centroid(item_embedding: torch.Tensor, item_per_dataset: int) -> List[int]:
    """distiller/item/utils.py:42-65: k-means (scikit-learn defaults), the member nearest to each centre."""
    import numpy as np
    from sklearn.cluster import KMeans

    assert isinstance(item_embedding, torch.Tensor) and item_embedding.ndim == 2, "item_embedding must be a 2-dimensional torch.Tensor"
    assert 1 <= item_per_dataset <= item_embedding.shape[0], "item_per_dataset must be between 1 and N"
    X = item_embedding.cpu().detach().numpy()
    km = KMeans(n_clusters=item_per_dataset).fit(X)
    picked = []
    for c, centre in enumerate(km.cluster_centers_):
        members = np.where(km.labels_ == c)[0]
        assert members.size > 0, f"Cluster {c} has no members"
        picked.append(int(members[int(np.argmin(np.linalg.norm(X[members] - centre, axis=1)))]))
    assert len(set(picked)) == item_per_dataset, "Duplicate indices found"
    return picked


class DistillSequenceCollator:
    """collator/distiller/collator.py:42-91: (dataset index, (sequence id, items)) samples -> BatchDistillationSequence."""

    def __init__(self, tokenizer, item_texts: List[Dict[int, str]], max_seq_len: int, separator: str = "; ", sequence_prompt: str = "",
                 reverse_sequence: bool = True):
        self.tokenizer, self.item_texts, self.max_seq_len = tokenizer, item_texts, max_seq_len
        self.separator, self.sequence_prompt, self.reverse_sequence = separator, sequence_prompt, reverse_sequence

    def __call__(self, batch):
        from .model_batch import BatchDistillationSequence

        ds_idx, seq_ids, texts = [], [], []
        for d, (sid, seq) in batch:
            if self.reverse_sequence:  # the last entry is the held-out target (or the -1 marker of item pseudo-sequences)
                seq = seq[:-1][::-1]
            ds_idx.append(d)
            seq_ids.append(sid)
            texts.append(self.sequence_prompt + self.separator.join(self.item_texts[d][i] for i in seq))
        enc = _encode_texts(self, texts)
        return BatchDistillationSequence(dataset_indexes=ds_idx, sequence_ids=torch.tensor(seq_ids, device=torch.device("cpu")), sequence=enc)


class DistillSequenceDataModule:
    """datamodule/distiller/sequence/datamodule.py:20-184: per domain an item dataloader (catalog encoding) and the pseudo-user
    sequences whose teacher rows are ``sequence_embeddings[d] @ item_embeddings[d].T``; training batches mix all domains."""

    def __init__(self, dataset_paths, tokenizer, batch_size: int, max_seq_len: int, max_attribute_len: int, max_items: int,
                 sequence_embeddings: List[torch.Tensor], train_data_split: str, sequence_per_dataset: Optional[int] = None, num_workers: int = 0,
                 item_prompt: Optional[str] = None, sequence_prompt: Optional[str] = None, valid_ratio: Optional[float] = None,
                 reverse_sequence: bool = True, num_sequences_per_dataset: Optional[int] = None, sample_method: str = "random"):
        assert valid_ratio is None or 0 <= valid_ratio <= 1, "valid_ratio must be between 0 and 1 or None"
        assert len(dataset_paths) == len(sequence_embeddings), "dataset_paths and user_embeddings must have the same length"
        self.dataset_paths, self.tokenizer = [Path(p) for p in dataset_paths], tokenizer
        self.batch_size, self.max_seq_len, self.max_attribute_len, self.max_items = batch_size, max_seq_len, max_attribute_len, max_items
        self.sequence_embeddings, self.train_data_split, self.sequence_per_dataset = sequence_embeddings, train_data_split, sequence_per_dataset
        self.num_workers, self.valid_ratio, self.reverse_sequence = num_workers, valid_ratio, reverse_sequence
        self.num_sequences_per_dataset, self.sample_method = num_sequences_per_dataset, sample_method
        self.item_prompt, self.sequence_prompt = item_prompt or "", sequence_prompt or ""
        self.score_train_datasets, self.score_valid_datasets = [], []
        self.item_datasets, self.item_collators, self.item_dataloaders, self.item_texts = [], [], [], []
        self.distill_collator = None

    _flatten_key_value = RecDataModule._flatten_key_value

    def setup(self, stage: str = "fit"):
        from torch.utils.data import Subset

        for path, seq_emb in zip(self.dataset_paths, self.sequence_embeddings):
            item_dataset, train, val, test, metadata, _, _ = load_json_files(path, self.max_items)
            if self.train_data_split == "item":
                dataset = RecItemAsSequenceDataset(item_dataset.items)
            elif self.train_data_split in ("train", "val", "test"):
                dataset = dict(train=train, val=val, test=test)[self.train_data_split]
            else:
                raise ValueError(f"Unknown train_data_split: {self.train_data_split}")
            assert len(dataset) == len(seq_emb), "item_dataset and sequence_embedding must have the same length"
            if self.num_sequences_per_dataset is not None:
                print(f"Sampling {self.num_sequences_per_dataset} sequences per dataset from {len(dataset)} total sequences")
                if self.sample_method == "random":
                    indices = torch.randperm(len(dataset))[: self.num_sequences_per_dataset].tolist()
                elif self.sample_method == "centroid":
                    indices = sample_centroid(seq_emb, self.num_sequences_per_dataset)
                elif self.sample_method == "popular":
                    indices = sample_popular(test.sequence, self.num_sequences_per_dataset)
                else:
                    raise ValueError(f"Unknown sample_method: {self.sample_method}")
                dataset = Subset(dataset, indices)
            item_text, collator = self._item_side(metadata)
            tr, va = split_sequences(dataset, self.valid_ratio)
            self.item_datasets.append(item_dataset)
            self.item_collators.append(collator)
            self.item_dataloaders.append(DataLoader(item_dataset, batch_size=self.batch_size, collate_fn=collator, shuffle=False,
                                                    num_workers=self.num_workers))
            self.score_train_datasets.append(tr)
            self.score_valid_datasets.append(va)
            self.item_texts.append(item_text)
        self.distill_collator = self._distill_collator()

    def _item_side(self, metadata):
        """per domain: the item representation the collators index by item id, and the catalog collator"""
        item_text = {i: self._flatten_key_value(m) for i, m in metadata.items()}
        return item_text, SingleItemCollator(self.tokenizer, item_text, self.max_seq_len, self.item_prompt)

    def _distill_collator(self):
        return DistillSequenceCollator(self.tokenizer, self.item_texts, self.max_seq_len, "; ", self.sequence_prompt, self.reverse_sequence)

    def train_dataloader(self):
        return DataLoader(ChainedDataset(self.score_train_datasets), batch_size=self.batch_size, collate_fn=self.distill_collator,
                          num_workers=self.num_workers, shuffle=True)

    def val_dataloader(self):
        return [DataLoader(ChainedDataset([v], start_dataset_idx=i), batch_size=self.batch_size, collate_fn=self.distill_collator,
                           num_workers=self.num_workers, shuffle=False)
                for i, v in enumerate(self.score_valid_datasets) if v is not None]


class RecformerDistillSequenceCollator:
    """collator/distiller/recformer.py:47-82: pseudo-user sequences from pre-tokenised items, all but the last entry, NOT
    reversed (unlike the evaluation collator)."""

    def __init__(self, bos_token_id: int, pad_token_id: int, tokenized_items: List[Dict[int, TokenizedItem]], max_seq_len: int):
        self.bos_token_id, self.pad_token_id, self.tokenized_items, self.max_seq_len = bos_token_id, pad_token_id, tokenized_items, max_seq_len

    def __call__(self, batch):
        from .model_batch import BatchDistillationSequence

        ds_idx, seq_ids, seqs = [], [], []
        for d, (sid, seq) in batch:
            ds_idx.append(d)
            seq_ids.append(sid)
            seqs.append(concat_tokenized_items([self.tokenized_items[d][i] for i in seq[:-1]], self.bos_token_id))
        return BatchDistillationSequence(dataset_indexes=ds_idx, sequence_ids=seq_ids,
                                         sequence=pad_tokenized_sequences(seqs, self.pad_token_id, self.max_seq_len))


class DistillSequenceDataModuleForRecformer(DistillSequenceDataModule):
    """datamodule/distiller/sequence/recformer.py:28-180: the same sampling / splitting with pre-tokenised items (one attribute-id
    map shared by all domains) and the Recformer collators."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.bos_token_id, self.pad_token_id = self.tokenizer.bos_token_id, self.tokenizer.pad_token_id
        self._attr_name_id_map = _FirstSeenIds()
        self.tokenized_items = self.item_texts  # filled by setup(): per domain, item id -> TokenizedItem

    def _item_side(self, metadata):
        tok = {i: tokenize_item(metadata[i], self.tokenizer, self._attr_name_id_map, self.max_attribute_len) for i in metadata}
        return tok, RecformerSingleItemCollator(self.bos_token_id, self.pad_token_id, tok, self.max_seq_len)

    def _distill_collator(self):
        return RecformerDistillSequenceCollator(self.bos_token_id, self.pad_token_id, self.tokenized_items, self.max_seq_len)
