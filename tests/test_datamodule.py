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
