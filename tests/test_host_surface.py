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
