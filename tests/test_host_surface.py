"""Host-side pieces of the drop-in surface that need no GPU: argument lists, batch relocation metadata, the alpha-file callbacks."""
import inspect
from types import SimpleNamespace

import pytest
import torch


def test_test_model_has_the_reference_argument_list():
    """utils.py:32-51 of the reference: names, order and defaults -- merge_test.py:91-110 calls it by keyword."""
    from mergerec_amd.utils import test_model

    params = list(inspect.signature(test_model).parameters.items())
    assert [n for n, _ in params] == [
        "module", "model_type", "data_paths", "model_tokenizer", "batch_size", "max_seq_len", "max_attribute_len", "max_items", "num_workers",
        "sequence_prompt", "item_prompt", "reverse_sequence", "precision", "data_split", "metrics_path", "predictions_path",
        "item_embeddings_path", "user_embeddings_path"]
    assert all(p.default is inspect.Parameter.empty for _, p in params[:14]) and all(p.default is None for _, p in params[14:])


def test_load_merging_module_keeps_the_reference_arguments_in_front():
    from mergerec_amd.merger import load_merging_module

    names = list(inspect.signature(load_merging_module).parameters)
    assert names[:11] == ["merge_type", "learn_type", "model", "pretrain_state_dict", "finetune_state_dicts", "ignore_keys", "ties_density",
                          "initial_global_weight", "initial_global_bias", "initial_per_weight", "disable_softmax"]  # _factory.py:27-39


def test_to_device_records_host_lengths():
    from mergerec_amd.model_batch import BatchItem, BatchSequence, Encoding

    mask = torch.tensor([[1, 1, 1, 0], [1, 0, 0, 0], [1, 1, 1, 1]])
    ids = torch.arange(12).view(3, 4)
    b = BatchSequence(sequence={"input_ids": ids, "attention_mask": mask}, labels=torch.tensor([1, 2, 3])).to("cpu")
    assert isinstance(b.sequence, Encoding) and b.sequence.host_lens.tolist() == [3, 1, 4]
    assert torch.equal(b.sequence["input_ids"], ids) and "labels" not in b.sequence
    again = b.to("cpu")  # a second move keeps the lengths taken the first time
    assert again.sequence.host_lens.tolist() == [3, 1, 4]
    from transformers import BatchEncoding

    it = BatchItem(items=BatchEncoding({"input_ids": ids, "attention_mask": mask})).to("cpu")
    assert it.items.host_lens.tolist() == [3, 1, 4] and set(it.items) == {"input_ids", "attention_mask"}


def test_alpha_file_round_trip_and_best_alpha(tmp_path):
    from mergerec_amd.module.callbacks import SaveWeightsCallback, WeightCheckpointCallback
    from mergerec_amd.utils import load_alpha_file

    weights = [{"global_weights": {"all": [1.0]}, "global_biases": {"all": [0.0]}, "per_weights": {"all": [0.1 * s, 0.2]}} for s in range(6)]
    module = SimpleNamespace(merged_model=SimpleNamespace(serialize_weights=lambda: weights[trainer.global_step]))
    trainer = SimpleNamespace(current_epoch=0, global_step=0, callback_metrics={})
    cb = SaveWeightsCallback(version="run", save_dir=tmp_path / "w", log_every_steps=2)
    for step in range(6):
        trainer.global_step = step
        cb.on_train_batch_end(trainer, module, None, None, batch_idx=step)
    cb.on_train_epoch_end(trainer, module)
    cb.teardown(trainer, module, "fit")
    cb.teardown(trainer, module, "fit")  # idempotent
    lines = (tmp_path / "w" / "run.jsonl").read_text().strip().splitlines()
    assert len(lines) == 3 and lines[1].startswith("{'epoch': 0, 'step': 2, 'weights': {")  # str(dict), one per logged step
    assert load_alpha_file(tmp_path / "w" / "run.jsonl", 2) == weights[4]

    ck = WeightCheckpointCallback(monitor=r"val/loss_epoch/dataloader_idx_\d+")
    loaded = []
    module.merged_model.load_weights_from_dict = loaded.append
    ck.load_weights(module)
    assert loaded == []
    for step, (a, b) in enumerate([(2.0, 4.0), (1.0, 2.0), (3.0, 0.5)]):
        trainer.global_step = step
        trainer.callback_metrics = {"val/loss_epoch/dataloader_idx_0": torch.tensor(a), "val/loss_epoch/dataloader_idx_1": torch.tensor(b),
                                    "val/loss_epoch": torch.tensor(-100.0)}
        ck.on_validation_epoch_end(trainer, module)
    assert ck.best_score == 1.5 and ck.best_weights == weights[1]
    ck.load_weights(module)
    assert loaded == [weights[1]]
    trainer.callback_metrics = {"train/loss": torch.tensor(1.0)}
    with pytest.raises(RuntimeError):
        ck.on_validation_epoch_end(trainer, module)


def test_coalesced_batches_keep_each_rows_own_padded_width():
    """pooling_method="mean" averages over the padded width of the batch a sequence came in (encoder/_base.py:42-43 on upstream's own
    batches): when the evaluation loop merges batches of different widths into one kernel pass, every row keeps its width
    (Encoding.host_pad_len), through a second merge and through .to(device)"""
    import torch

    from mergerec_amd.data import coalesce_batches
    from mergerec_amd.model_batch import BatchItem, BatchSequence

    def enc(rows, width, fill):
        ids = torch.full((rows, width), 1, dtype=torch.int64)
        mask = torch.zeros(rows, width, dtype=torch.int64)
        ids[:, :fill], mask[:, :fill] = 5, 1
        return {"input_ids": ids, "attention_mask": mask}

    merged = list(coalesce_batches([BatchItem(items=enc(2, 9, 4)), BatchItem(items=enc(3, 23, 20)), BatchItem(items=enc(1, 5, 5))], max_tokens=10 ** 6))
    assert len(merged) == 1 and merged[0].items["input_ids"].shape == (6, 23)
    assert merged[0].items.host_pad_len.tolist() == [9, 9, 23, 23, 23, 5]
    moved = merged[0].to("cpu")
    assert moved.items.host_pad_len.tolist() == [9, 9, 23, 23, 23, 5] and moved.items.host_lens.tolist() == [4, 4, 20, 20, 20, 5]
    # equal widths: nothing to carry (the packer then uses the tensor's own width)
    same = list(coalesce_batches([BatchItem(items=enc(2, 9, 4)), BatchItem(items=enc(3, 9, 9))], max_tokens=10 ** 6))
    assert getattr(same[0].items, "host_pad_len", None) is None
    # sequences with labels go the same way; an already merged batch merged again keeps its rows' widths
    seqs = list(coalesce_batches([BatchSequence(sequence=merged[0].items, labels=torch.zeros(6, dtype=torch.int64)),
                                  BatchSequence(sequence=enc(2, 30, 30), labels=torch.ones(2, dtype=torch.int64))], max_tokens=10 ** 6))
    assert seqs[0].sequence.host_pad_len.tolist() == [9, 9, 23, 23, 23, 5, 30, 30] and seqs[0].labels.tolist() == [0] * 6 + [1, 1]


def test_distillation_log_keeps_tensors_until_read():
    """DistillSequenceModule.log stores the loss TENSOR (a float() there is a device -> host wait inside every training step); readers of
    ``logged`` get floats"""
    import torch

    from mergerec_amd.module.distiller import _LazyLog

    log = _LazyLog()
    t = torch.tensor(2.5)
    log["train/loss"] = t
    log["lr"] = 1e-3
    assert dict.__getitem__(log, "train/loss") is t            # kept as given
    assert log["train/loss"] == 2.5 and isinstance(log["train/loss"], float)
    assert log.get("missing") is None and log.get("lr") == 1e-3
    assert sorted(log.items()) == [("lr", 1e-3), ("train/loss", 2.5)] and sorted(log.values()) == [1e-3, 2.5]
