#!/usr/bin/env python3
"""Drop-in for the reference's finetune_test.py (finetune_test.py:11-57): ONE fine-tuned checkpoint, no merge -- the single-model
path of BASELINE config 1.  Same flag names (TestSingleConfig, configs/test.py:21-30), argparse instead of tyro.

  python finetune_test.py --model_type BLAIR_BASE --finetune_checkpoint_path ckpt/state_dict.pt \\
      --data_path datasets/Pantry --tokenizer_path /path/to/roberta-base --data_split test

``--finetune_checkpoint_path`` is the ``state_dict.pt`` that scripts/extract.py writes (keys ``model.model.*`` plus
``item_embeddings``, which is dropped: the catalog is re-encoded).  ``--data_path`` is a dataset directory in the reference's JSON
format (needs ``--tokenizer_path``) or a spec of mergerec_amd/data.py (``synthetic:Name:M:U``, a pre-tokenised .pt)."""
from __future__ import annotations

import os
import sys
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(argv=None):
    import merge_test as mt

    # TestSingleConfig = the merge CLI's flags minus the merge ones, with singular data / checkpoint paths
    argv = list(sys.argv[1:] if argv is None else argv)
    single = {"--data_path": "--data_paths", "--finetune_checkpoint_path": "--finetune_checkpoint_paths", "--data_split": "--test_data_split"}
    argv = [single.get(a, a) for a in argv]
    config = mt.parse(argv + ["--merge_type", "TASK_VECTOR", "--learn_type", "TASK_WISE"])
    if len(config.data_paths) != 1 or len(config.finetune_checkpoint_paths) != 1:
        raise SystemExit("finetune_test.py takes exactly one --data_path and one --finetune_checkpoint_path")

    from mergerec_amd import parallel
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.utils import remove_duplicate_prefix, test_model

    rank, _ = parallel.init_from_env()
    torch.manual_seed(config.seed)
    model = ModelType[config.model_type].value(
        model_name_or_path=config.model_path, tokenizer_name_or_path=config.tokenizer_path, lora_config=None,
        pooling_method=config.pooling_method, model_kwargs=dict(config.model_kwargs), tokenizer_kwargs=dict(config.tokenizer_kwargs),
    )
    module = RecModule(model=model, evaluator=Evaluator(metrics=config.metric_names, ks=config.ks), negative_sample=None, similarity=config.similarity)

    ckpt = config.finetune_checkpoint_paths[0]
    if not str(ckpt).startswith("synthetic:"):  # finetune_test.py:31-35
        sd = torch.load(ckpt, map_location="cpu")
        sd.pop("item_embeddings")
        model.load_state_dict(remove_duplicate_prefix(sd))

    path = config.data_paths[0]
    tokenizer = model.tokenizer
    if (Path(path) / "train.json").exists():
        from mergerec_amd.datamodule import load_tokenizer

        if not config.tokenizer_path:
            raise SystemExit("--tokenizer_path <local tokenizer directory> is required for JSON dataset directories (the box is offline)")
        tokenizer = load_tokenizer(config.tokenizer_path)
    # finetune_test.py:36-55
    _, metrics, _, _ = test_model(
        module=module, model_type=ModelType[config.model_type], data_paths=[path], model_tokenizer=tokenizer, batch_size=config.batch_size,
        max_seq_len=config.max_seq_len, max_attribute_len=config.max_attribute_len, max_items=config.max_items, num_workers=config.num_workers,
        sequence_prompt=config.sequence_prompt, item_prompt=config.item_prompt,
        reverse_sequence=str(config.reverse_sequence).lower() in ("1", "true", "yes"), precision=config.precision,
        data_split=config.test_data_split, metrics_path=config.metrics_path, predictions_path=config.predictions_path,
        item_embeddings_path=config.item_embeddings_path, user_embeddings_path=config.user_embeddings_path)
    name = str(path).split(":")[1] if str(path).startswith("synthetic:") else Path(path).name
    if rank == 0:
        print(name, {k: round(v, 5) for k, v in metrics[0].items()})
    return metrics


if __name__ == "__main__":
    main()
