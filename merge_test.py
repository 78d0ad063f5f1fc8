#!/usr/bin/env python3
"""Drop-in for the reference's merge_test.py (merge_test.py:16-110) on MI355X: same flag names (TestMergeConfig,
configs/base.py:22-108, configs/test.py:34-43), argparse instead of tyro.  Flags of subsystems outside the path
(--lora.*) are accepted and ignored with a note; --data_paths may be dataset directories in the reference's JSON format
(then --tokenizer_path must be a local tokenizer directory) or the pre-tokenised / synthetic specs of mergerec_amd/data.py;
`python -m torch.distributed.run --nproc-per-node N merge_test.py ...` shards the run over N GPUs (same outputs); --precision 32-true keeps the model's arithmetic (default bf16x6, fp32-grade); bf16-mixed (the reference's default) selects bf16x3.

Example (synthetic weights + data, 2-domain merge):
  python merge_test.py --model_type BLAIR_BASE --model_kwargs init_seed 7 \
      --finetune_checkpoint_paths synthetic:1 synthetic:2 --merge_type TASK_VECTOR --learn_type TASK_WISE \
      --weight_file average --data_paths synthetic:Pantry --test_data_split test --train_data_split item
"""
from __future__ import annotations

import argparse
import os
import sys
from collections import OrderedDict
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _kv(pairs):
    out = {}
    it = iter(pairs or [])
    for k in it:
        v = next(it)
        for cast in (int, float):
            try:
                v = cast(v)
                break
            except ValueError:
                continue
        out[k] = v
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
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
    ap.add_argument("--precision", default="32-true")
    ap.add_argument("--num_workers", type=int, default=0)
    ap.add_argument("--metric_names", nargs="+", default=["NDCG", "RECALL"])
    ap.add_argument("--ks", nargs="+", type=int, default=[1, 5, 10, 50])
    ap.add_argument("--model_kwargs", nargs="*", default=[])
    ap.add_argument("--tokenizer_kwargs", nargs="*", default=[])
    ap.add_argument("--data_paths", nargs="+", required=True)
    ap.add_argument("--test_data_paths", nargs="*", default=[])
    ap.add_argument("--finetune_checkpoint_paths", nargs="+", required=True)
    ap.add_argument("--train_data_split", default="item")
    ap.add_argument("--test_data_split", choices=["val", "test"], default="test")
    ap.add_argument("--data_split", choices=["val", "test"], default=None)  # deprecated alias (configs/base.py:80-100)
    ap.add_argument("--merge_type", required=True)
    ap.add_argument("--learn_type", required=True)
    ap.add_argument("--ties_density", type=float, default=0.2)
    ap.add_argument("--use_softmax", action="store_true")
    ap.add_argument("--weight_file", default=None)
    ap.add_argument("--weight_file_line", default=None)
    ap.add_argument("--metrics_path", default=None)
    ap.add_argument("--predictions_path", default=None)
    ap.add_argument("--item_embeddings_path", default=None)
    ap.add_argument("--user_embeddings_path", default=None)
    cfg, unknown = ap.parse_known_args(argv)
    skip = False  # True only for the ONE token that directly follows a --lora* flag (its value)
    for u in unknown:
        if u.startswith("--lora"):
            print(f"note: {u} ignored (LoRA wrappers are outside the merged-inference path)")
            skip = "=" not in u  # `--lora.r 8`: the value follows; `--lora.r=8`: nothing to skip
        elif u.startswith("--") or not skip:
            ap.error(f"unrecognized argument {u}")
        else:
            skip = False
    if cfg.data_split is not None:  # configs/base.py:91-101: the alias never overrides test_data_split
        import warnings

        if cfg.test_data_split != cfg.data_split:
            warnings.warn("data_split is set but does not match test_data_split. Using test_data_split for merging.", UserWarning)
        else:
            warnings.warn("data_split is deprecated and will be removed in future versions. Use test_data_split instead.", DeprecationWarning)
    cfg.model_type = cfg.model_type.upper()
    cfg.merge_type, cfg.learn_type = cfg.merge_type.upper(), cfg.learn_type.upper()
    cfg.metric_names = [m.upper() for m in cfg.metric_names]
    cfg.model_kwargs, cfg.tokenizer_kwargs = _kv(cfg.model_kwargs), _kv(cfg.tokenizer_kwargs)
    if not cfg.test_data_paths:
        cfg.test_data_paths = cfg.data_paths
    return cfg


def main(argv=None):
    config = parse(argv)
    from mergerec_amd import parallel
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.utils import load_alpha_file, remove_duplicate_prefix, test_model

    # one process per GPU under `python -m torch.distributed.run --nproc-per-node N merge_test.py ...`: the arena slices, the catalog rows
    # and the user sequences are dealt over the ranks (RCCL all-gathers); a plain `python merge_test.py ...` is the single-GPU run
    rank, world_size = parallel.init_from_env()
    torch.manual_seed(config.seed)
    model = ModelType[config.model_type].value(
        model_name_or_path=config.model_path, tokenizer_name_or_path=config.tokenizer_path, lora_config=None,
        pooling_method=config.pooling_method, model_kwargs=dict(config.model_kwargs), tokenizer_kwargs=dict(config.tokenizer_kwargs),
    )
    pretrain = OrderedDict((k, v.cpu().clone()) for k, v in model.state_dict().items())

    finetune_state_dicts = []
    for path in config.finetune_checkpoint_paths:  # merge_test.py:19-25
        if str(path).startswith("synthetic:"):
            g = torch.Generator().manual_seed(1000 + int(str(path).split(":")[1]))
            sd = OrderedDict((k, v if k.endswith("position_ids") else v + 1e-3 * torch.randn(v.shape, generator=g)) for k, v in pretrain.items())
        else:
            sd = torch.load(path, map_location="cpu")
            sd.pop("item_embeddings", None)
            sd = remove_duplicate_prefix(sd)
        finetune_state_dicts.append(sd)

    merged_model = load_merging_module(
        merge_type=MergeType[config.merge_type], learn_type=LearnType[config.learn_type], model=model, pretrain_state_dict=pretrain,
        finetune_state_dicts=finetune_state_dicts, ignore_keys=set(), ties_density=config.ties_density, disable_softmax=not config.use_softmax,
    )
    n = len(finetune_state_dicts)
    groups = list(merged_model.per_weights.keys())
    wf = Path(config.weight_file).name if config.weight_file else "average"
    if wf == "average":  # merge_test.py:47-55
        per = [1.0 / n] * n
        weights = {"global_weights": {g: [1.0] for g in groups}, "global_biases": {g: [0.0] for g in groups}, "per_weights": {g: per for g in groups}}
        print(f"Using average weights for {n} models.")
    elif wf == "uniform":  # :56-65
        w = float(config.weight_file_line)
        weights = {"global_weights": {g: [1.0] for g in groups}, "global_biases": {g: [0.0] for g in groups}, "per_weights": {g: [w] * n for g in groups}}
        print(f"Using uniform weights for {n} models: {w}.")
    else:  # :67-68 (literal_eval instead of eval)
        weights = load_alpha_file(config.weight_file, int(config.weight_file_line))
    merged_model.load_weights_from_dict(weights)

    state_dict = {k: v.detach() for k, v in merged_model.get_state_dict().items()}
    model = ModelType[config.model_type].value(
        model_name_or_path=config.model_path, tokenizer_name_or_path=config.tokenizer_path, lora_config=None,
        pooling_method=config.pooling_method, model_kwargs=dict(config.model_kwargs), tokenizer_kwargs=dict(config.tokenizer_kwargs),
    )
    model.load_state_dict(state_dict)
    module = RecModule(model=model, evaluator=Evaluator(metrics=config.metric_names, ks=config.ks), negative_sample=None, similarity=config.similarity)

    # merge_test.py:91-110, argument for argument
    _, metrics, scores, labels = test_model(
        module=module,
        model_type=ModelType[config.model_type],
        data_paths=config.test_data_paths,
        model_tokenizer=model.tokenizer,
        batch_size=config.batch_size,
        max_seq_len=config.max_seq_len,
        max_attribute_len=config.max_attribute_len,
        max_items=config.max_items,
        num_workers=config.num_workers,
        sequence_prompt=config.sequence_prompt,
        item_prompt=config.item_prompt,
        reverse_sequence=str(config.reverse_sequence).lower() in ("1", "true", "yes"),
        precision=config.precision,
        data_split=config.test_data_split,
        metrics_path=config.metrics_path,
        predictions_path=config.predictions_path,
        item_embeddings_path=config.item_embeddings_path,
        user_embeddings_path=config.user_embeddings_path,
    )
    if rank == 0:
        for p, m in zip(config.test_data_paths, metrics):
            print(str(p).split(":")[1] if str(p).startswith("synthetic:") else Path(p).name, {k: round(v, 5) for k, v in m.items()})
    return metrics


if __name__ == "__main__":
    main()
