"""Input-pipeline mirrors (mergerec_amd/datamodule.py) against what the REFERENCE's datamodules / collators produced on the
synthetic domain + local tokenizer fixtures (tests/golden/g8_datamodule.pt, oracle/gen_golden_datamodule.py)."""
import pytest
import torch

from tests.conftest import GOLDEN, load_golden


@pytest.fixture(scope="module")
def tok():
    from mergerec_amd.datamodule import load_tokenizer

    return load_tokenizer(GOLDEN / "mini_tokenizer")


def _same_encoding(got, want):
    assert set(got.keys()) == set(want.keys()), (got.keys(), want.keys())
    for k in want:
        assert got[k].dtype == want[k].dtype and torch.equal(got[k], want[k]), k


@pytest.mark.parametrize("name,kwargs", [("blair_reversed", dict(reverse_sequence=True)),
                                         ("blair_forward", dict(reverse_sequence=False, sequence_prompt="History: ", item_prompt="Item: "))])
def test_text_datamodule_matches_reference(tok, name, kwargs):
    from mergerec_amd.datamodule import RecDataModule

    g = load_golden("g8_datamodule.pt")[name]
    dm = RecDataModule(GOLDEN / "mini_dataset", tok, batch_size=8, max_seq_len=96, max_attribute_len=12, max_items=20, **kwargs)
    dm.setup("test")
    assert dm.item_text == g["item_text"]
    assert (len(dm.item_dataset), len(dm.val_dataset), len(dm.test_dataset)) == (g["n_items"], g["n_val"], g["n_test"])
    assert [dm.test_dataset[i] for i in range(len(dm.test_dataset))] == g["test_sequences"]
    assert [dm.val_dataset[i] for i in range(5)] == g["val_sequences"]
    for b, want in zip(list(dm.item_dataloader())[:3], g["item_batches"]):
        _same_encoding(b.items, want)
    for split, dl in (("val", dm.val_dataloader()), ("test", dm.test_dataloader())):
        bs = list(dl)
        for b, want in zip(bs[:2] + bs[-1:], g[f"{split}_batches"]):
            _same_encoding(b.sequence, want["sequence"])
            assert torch.equal(b.labels, want["labels"])


def test_recformer_datamodule_matches_reference(tok):
    from mergerec_amd.datamodule import RecDataModuleForRecformer

    g = load_golden("g8_datamodule.pt")["recformer"]
    dm = RecDataModuleForRecformer(GOLDEN / "mini_dataset", tok, batch_size=8, max_seq_len=128, max_attribute_len=10, max_items=20)
    dm.setup("test")
    assert dict(dm._attr_name_id_map) == g["attr_ids"]
    assert {k: tuple(list(x) for x in v) for k, v in dm.tokenized_items.items()} == g["tokenized_items"]
    for b, want in zip(list(dm.item_dataloader())[:3], g["item_batches"]):
        _same_encoding(b.items, want)
    bs = list(dm.test_dataloader())
    for b, want in zip(bs[:2] + bs[-1:], g["test_batches"]):
        _same_encoding(b.sequence, want["sequence"])
        assert torch.equal(b.labels, want["labels"])


def test_pad_to_multiple_and_errors(tok, tmp_path):
    from mergerec_amd.datamodule import TokenizedSequence, load_tokenizer, pad_tokenized_sequences

    s = TokenizedSequence([0, 5, 6], [0, 1, 2], [0, 1, 1], [0, 1, 1], [1, 1, 1], [1, 0, 0])
    enc = pad_tokenized_sequences([s], pad_token_id=1, max_length=64, pad_to_multiple_of=8)
    assert enc["input_ids"].shape == (1, 8) and enc["input_ids"][0, 3:].tolist() == [1] * 5 and enc["token_type_ids"][0, 3:].tolist() == [3] * 5
    assert pad_tokenized_sequences([s], 1, max_length=2)["input_ids"].tolist() == [[0, 5]]
    with pytest.raises(FileNotFoundError):
        load_tokenizer(tmp_path / "nope")


@pytest.mark.parametrize("name,kind,split,n_seq,kw", [
    ("distill_text_item", "text", "item", 60, dict(sequence_prompt="Seq: ")),
    ("distill_text_test", "text", "test", 40, dict()),
    ("distill_recformer_item", "recformer", "item", 60, dict()),
    ("distill_recformer_val", "recformer", "val", 40, dict()),
])
def test_distill_datamodules_match_reference(tok, name, kind, split, n_seq, kw):
    """merge_train.py's data: same sampled / split pseudo-user sets (same torch RNG draws), same chained samples, same collated
    batches as the reference's DistillSequenceDataModule / ...ForRecformer"""
    from mergerec_amd.datamodule import DistillSequenceDataModule, DistillSequenceDataModuleForRecformer

    g = load_golden("g8_datamodule.pt")[name]
    cls = DistillSequenceDataModuleForRecformer if kind == "recformer" else DistillSequenceDataModule
    torch.manual_seed(123)
    root = GOLDEN / "mini_dataset"
    dm = cls([root, root], tok, batch_size=8, max_seq_len=96, max_attribute_len=12, max_items=20,
             sequence_embeddings=[torch.zeros(n_seq, 4), torch.zeros(n_seq, 4)], train_data_split=split, valid_ratio=0.25,
             num_sequences_per_dataset=30, sample_method="random", **kw)
    dm.setup("fit")
    chained = dm.train_dataloader().dataset
    assert len(chained) == g["n_train"]
    samples = [chained[i] for i in range(len(chained))]
    assert [(d, (int(sid), list(seq))) for d, (sid, seq) in samples] == g["samples"]
    for start, want in zip((0, 8, len(samples) - 5), g["batches"]):
        b = dm.distill_collator(samples[start:start + 8])
        assert list(b.dataset_indexes) == want["dataset_indexes"] and [int(x) for x in b.sequence_ids] == want["sequence_ids"]
        _same_encoding(b.sequence, want["sequence"])
    vals = [b for dl in dm.val_dataloader() for b in list(dl)[:1]]
    assert len(vals) == len(g["val_first"])
    for b, want in zip(vals, g["val_first"]):
        assert list(b.dataset_indexes) == want["dataset_indexes"] and [int(x) for x in b.sequence_ids] == want["sequence_ids"]
        _same_encoding(b.sequence, want["sequence"])
    for dl, want in zip(dm.item_dataloaders, g["item_batch0"]):
        _same_encoding(next(iter(dl)).items, want)


def test_sample_popular_and_chained_dataset():
    from mergerec_amd.datamodule import ChainedDataset, sample_popular

    g = load_golden("g8_datamodule.pt")["sample_popular"]
    assert sample_popular(g["sequences"], 7) == g["top"]
    c = ChainedDataset([[10, 11], [20], [30, 31, 32]], start_dataset_idx=5)
    assert len(c) == 6 and [c[i] for i in range(6)] == [(5, 10), (5, 11), (6, 20), (7, 30), (7, 31), (7, 32)] and c[-1] == (7, 32)
    with pytest.raises(ValueError):
        c[-7]
