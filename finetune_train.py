#!/usr/bin/env python3
"""Drop-in for the reference's finetune_train.py (finetune_train.py:51-118): fine-tune ONE encoder on ONE domain -- the producer
of the per-domain checkpoints that the merge consumes (recipes scripts/1_finetune/blair_base.sh, recformer_base.sh).  Same flag
names as FinetuneSingleConfig (configs/finetune.py:27-57 on top of configs/base.py:23-61), argparse instead of tyro, including the
dotted ``--negative_sample.in_batch`` / ``--negative_sample.k``; wandb / Lightning loggers are not part of the path.

  python finetune_train.py --model_type blair_base --model_path pretrained/state_dict.pt --tokenizer_path /path/to/roberta-base \\
      --batch_size 64 --negative_sample.in_batch --temperature 0.05 --warmup_steps 100 --data_path datasets/Arts --learning_rate 5e-5

Per micro-step: ONE packed encoder forward + backward over [sequences; targets] on the HIP training graph (exact fp32), in-batch
cross entropy at temperature 0.05; per optimizer step (every ``--gradient_accumulation_steps`` micro-steps): one fused
clip + AdamW launch over the parameter arena.  After every epoch: catalog re-encode + validation (val/NDCG@10 monitored: best
checkpoint kept, early stopping after ``--patience`` epochs without improvement); finally the test split with the best weights.
The checkpoint is what scripts/extract.py turns into ``state_dict.pt`` / ``item_embedding.pt``.

Under ``python -m torch.distributed.run --nproc-per-node N`` the batches of an epoch are dealt over the ranks and the gradient
arena is averaged in one all-reduce per optimizer step (data parallel; the reference is single-GPU)."""
from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path
from uuid import uuid4

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _number(text: str):
    """``--warmup_steps``: an int is a step count, a float a fraction of all optimizer steps (module.py:58-63)."""
    try:
        return int(text)
    except ValueError:
        return float(text)


def parse(argv=None):
    import merge_test as mt

    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # BaseConfig
    ap.add_argument("--model_type", required=True)
    ap.add_argument("--pooling_method", default="cls")
    ap.add_argument("--model_path", default=None)
    ap.add_argument("--tokenizer_path", default=None)
    ap.add_argument("--max_seq_len", type=int, default=512)
    ap.add_argument("--max_attribute_len", type=int, default=32)
    ap.add_argument("--max_items", type=int, default=50)
    ap.add_argument("--batch_size", type=int, default=32)
    ap.add_argument("--similarity", choices=["cosine", "dot"], default="cosine")
    ap.add_argument("--sequence_prompt", default=None)
    ap.add_argument("--item_prompt", default=None)
    ap.add_argument("--reverse_sequence", default="True")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--precision", default="bf16-mixed")
    ap.add_argument("--num_workers", type=int, default=0)
    ap.add_argument("--metric_names", nargs="+", default=["NDCG", "RECALL"])
    ap.add_argument("--ks", nargs="+", type=int, default=[1, 5, 10, 50])
    ap.add_argument("--model_kwargs", nargs="*", default=[])
    ap.add_argument("--tokenizer_kwargs", nargs="*", default=[])
    # FinetuneConfig / FinetuneSingleConfig
    ap.add_argument("--temperature", type=float, default=0.05)
    ap.add_argument("--patience", type=int, default=5)
    ap.add_argument("--learning_rate", type=float, default=5e-5)
    ap.add_argument("--max_epochs", type=int, default=100)
    ap.add_argument("--warmup_steps", type=_number, default=100)
    ap.add_argument("--weight_decay", type=float, default=0.0)
    ap.add_argument("--gradient_accumulation_steps", type=int, default=4)
    ap.add_argument("--negative_sample.k", dest="negative_k", type=int, default=None)
    ap.add_argument("--negative_sample.in_batch", dest="negative_in_batch", action="store_true")
    ap.add_argument("--gradient_clip_val", type=float, default=None)
    ap.add_argument("--log_every_n_steps", type=int, default=50)
    ap.add_argument("--valid_metric", default="val/NDCG@10")
    ap.add_argument("--data_path", required=True)
    # not in the reference: where checkpoints go (Lightning: <logger project>/<run id>/checkpoints) and a step cap for smoke runs
    ap.add_argument("--default_root_dir", default=None)
    ap.add_argument("--max_steps", type=int, default=None)
    cfg, unknown = ap.parse_known_args(argv)
    for u in unknown:
        if u.startswith("--lora"):
            print(f"note: {u} ignored (LoRA wrappers are not built; full fine-tuning)")
        elif u.startswith("--"):
            raise SystemExit(f"unknown flag {u}")
    cfg.model_type = cfg.model_type.upper()
    cfg.metric_names = [m.upper() for m in cfg.metric_names]
    cfg.model_kwargs, cfg.tokenizer_kwargs = mt._kv(cfg.model_kwargs), mt._kv(cfg.tokenizer_kwargs)
    return cfg


def main(argv=None):
    from merge_train import _init_distributed

    rank, world = _init_distributed()
    config = parse(argv)
    from mergerec_amd.configs import NegativeSampleConfig, NegativeSampleOption
    from mergerec_amd.datamodule import load_tokenizer
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.module.callbacks import ItemEncodingCallback, ItemEncodingNegativeSampleCallback
    from mergerec_amd.utils import FinetuneTrainer, get_data_module

    import random

    import numpy as np

    # L.seed_everything (python, numpy, torch).  With several ranks the data-side streams (the collators' random prefix cuts and
    # sampled negatives) are offset by the rank so the shards are not augmented identically; the shard assignment itself uses its own
    # shared seed (parallel.shard_indices), and the model initialisation below is seeded identically everywhere.
    random.seed(config.seed + rank)
    np.random.seed(config.seed + rank)
    torch.manual_seed(config.seed)
    config.model_kwargs.setdefault("dropout_seed", config.seed + rank)  # HF dropout in train() mode, mask keyed by --seed (csrc/dropout.h)
    if not config.tokenizer_path:
        raise SystemExit("--tokenizer_path <local tokenizer directory> is required (the box is offline)")
    negative_sample = NegativeSampleConfig(k=config.negative_k, in_batch=config.negative_in_batch)
    model_type = ModelType[config.model_type]
    model = model_type.value(model_name_or_path=config.model_path, tokenizer_name_or_path=config.tokenizer_path, lora_config=None,
                             pooling_method=config.pooling_method, model_kwargs=dict(config.model_kwargs),
                             tokenizer_kwargs=dict(config.tokenizer_kwargs))
    module = RecModule(model=model, evaluator=Evaluator(metrics=config.metric_names, ks=config.ks), temperature=config.temperature,
                       learning_rate=config.learning_rate, warmup_steps=config.warmup_steps, weight_decay=config.weight_decay,
                       negative_sample=negative_sample, similarity=config.similarity)
    reverse = str(config.reverse_sequence).lower() in ("1", "true", "yes")
    datamodule = get_data_module(model_type, config.batch_size, Path(config.data_path), config.item_prompt, config.max_attribute_len,
                                 config.max_items, config.max_seq_len, load_tokenizer(config.tokenizer_path), negative_sample,
                                 config.num_workers, reverse, config.sequence_prompt)
    datamodule.setup("fit")

    # finetune_train.py:85-89 picks the callback by the truthiness of the config OBJECT (always the negative-sample one); the intent --
    # full-catalog training needs the catalog at the start of every training epoch -- is followed here
    item_cb = (ItemEncodingCallback if negative_sample.mode == NegativeSampleOption.FULL else ItemEncodingNegativeSampleCallback)(
        datamodule.item_dataloader())
    root = Path(config.default_root_dir) if config.default_root_dir else Path("MergeRecFineTune") / str(uuid4())[:8]
    trainer = FinetuneTrainer(max_epochs=config.max_epochs, accumulate_grad_batches=config.gradient_accumulation_steps,
                              gradient_clip_val=config.gradient_clip_val, precision=config.precision, callbacks=[item_cb],
                              monitor=config.valid_metric, patience=config.patience, default_root_dir=root,
                              log_every_n_steps=config.log_every_n_steps, max_steps=config.max_steps, verbose=rank == 0)
    trainer.fit(module, datamodule)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()  # rank 0 wrote the best checkpoint
        best = [str(trainer.best_model_path) if trainer.best_model_path else None]
        dist.broadcast_object_list(best, src=0)
        trainer.best_model_path = Path(best[0]) if best[0] else None
    metrics = trainer.test(module, datamodule, ckpt_path="best")
    if rank == 0:
        print("best checkpoint:", trainer.best_model_path)
        print(Path(config.data_path).name, {k: round(v, 5) for k, v in metrics[0].items()})
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()
    return trainer, metrics


if __name__ == "__main__":
    main()
