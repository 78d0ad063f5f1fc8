#!/usr/bin/env python3
"""Drop-in for the reference's merge_train.py (merge_train.py:103-222): the collaborative-merging optimisation loop -- learn the
merging coefficients alpha by distilling every domain's fine-tuned model into the merged one (BASELINE config 5's loop; recipe
scripts/3_mergerec/blair_base_taskvector_taskwise.sh).  Same flag names as DistillSequenceConfig (configs/distill.py:8-67 on top
of the merge_test flags), argparse instead of tyro; wandb / Lightning loggers are not part of the path.

Per step: re-merge (differentiable) -> encode 16 pseudo-user sequences -> per-domain logits against the frozen catalog
embeddings -> fused distillation loss vs the teacher rows -> encoder backward -> d loss / d alpha -> Adam on alpha.  BLaIR
(RoBERTa) and Recformer (Longformer) models.

Teacher embeddings: ``--item_embeddings_paths`` / ``--sequence_embeddings_paths`` are the ``item_embedding.pt`` files of
scripts/extract.py, or the single word ``auto`` to encode every domain's catalog with its own fine-tuned checkpoint first.

  python merge_train.py --model_type BLAIR_BASE --model_kwargs init_seed 7 --finetune_checkpoint_paths synthetic:1 synthetic:2 \\
      --data_paths tests/golden/mini_dataset tests/golden/mini_dataset --tokenizer_path tests/golden/mini_tokenizer \\
      --item_embeddings_paths auto --sequence_embeddings_paths auto --train_data_split item --test_data_split test \\
      --merge_type task_vector --learn_type task_wise --loss_type SINGLE_PSEUDO_LABEL_KD --coefficient 1000 --max_steps 20
"""
from __future__ import annotations

import os
import sys
from collections import OrderedDict
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _pop(argv, flag, default=None, cast=str, many=False):
    """remove ``flag value...`` from argv and return the value(s) (flags merge_test's parser does not know)"""
    if flag not in argv:
        return default
    i = argv.index(flag)
    j = i + 1
    vals = []
    while j < len(argv) and not argv[j].startswith("--"):
        vals.append(argv[j])
        j += 1
        if not many:
            break
    del argv[i:j]
    if many:
        return [cast(v) for v in vals]
    return cast(vals[0]) if vals else default


def _init_distributed():
    """Under torch.distributed.run (one process per GPU): pin this process to its GPU BEFORE the HIP runtime starts (so every rank
    addresses its device as cuda:0) and join the process group (RCCL; ``MERGEREC_DIST_BACKEND=gloo`` + ``MERGEREC_SHARE_GPU=1`` for a
    rehearsal with several ranks on one GPU).  Data parallelism here = each rank trains on its own shard of every epoch's pseudo
    users and the ranks average d loss / d alpha (mergerec_amd.parallel.allreduce_mean_grads)."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return 0, 1
    if os.environ.get("MERGEREC_SHARE_GPU", "0") != "1":
        os.environ.setdefault("HIP_VISIBLE_DEVICES", os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist

    dist.init_process_group(os.environ.get("MERGEREC_DIST_BACKEND", "nccl"))
    return dist.get_rank(), dist.get_world_size()


def main(argv=None):
    import merge_test as mt

    rank, world = _init_distributed()

    argv = list(sys.argv[1:] if argv is None else argv)
    opt = dict(
        item_embeddings_paths=_pop(argv, "--item_embeddings_paths", [], many=True),
        sequence_embeddings_paths=_pop(argv, "--sequence_embeddings_paths", [], many=True),
        loss_type=_pop(argv, "--loss_type", "kd").upper(),
        temperature=_pop(argv, "--temperature", 0.05, float),
        coefficient=_pop(argv, "--coefficient", 1.0, float),
        learning_rate=_pop(argv, "--learning_rate", 1e-4, float),
        max_steps=_pop(argv, "--max_steps", None, int),
        max_epochs=_pop(argv, "--max_epochs", None, int),
        valid_ratio=_pop(argv, "--valid_ratio", None, float),
        initial_per_weight=_pop(argv, "--initial_per_weight", 0.2, float),
        num_sequences_per_dataset=_pop(argv, "--num_sequences_per_dataset", None, int),
        sample_method=_pop(argv, "--sample_method", "random"),
        weights_dir=_pop(argv, "--weights_dir", "weights"),
        skip_test=_pop(argv, "--skip_test", "false").lower() in ("1", "true", "yes"),
        result_path=_pop(argv, "--result_path", None),  # every rank saves its result dict to <result_path>.rank<r>.pt
        loss_fn_kwargs=mt._kv(_pop(argv, "--loss_fn_kwargs", [], many=True)),  # configs/distill.py:14,32
    )
    for unused in ("--patience", "--gradient_accumulation_steps"):  # declared by DistillConfig, read by nothing in merge_train.py
        _pop(argv, unused)
    config = mt.parse(argv)
    from mergerec_amd.datamodule import DistillSequenceDataModule, DistillSequenceDataModuleForRecformer, load_tokenizer
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.merger import LearnType, LossType, MergeType, load_merging_module
    from mergerec_amd.module import (DistillSequenceModule, ModelType, MultiDatasetItemEncodingCallback, RecModule, SaveWeightsCallback,
                                     distill_loss_factory, teacher_scores)
    from mergerec_amd.module.callbacks import ItemEncoderMixin, WeightCheckpointCallback
    from mergerec_amd.utils import DistillTrainer, remove_duplicate_prefix, test_model

    torch.manual_seed(config.seed)
    # the training graph applies HF's dropout (hidden / attention 0.1 unless --model_kwargs overrides them) in train() mode; its
    # counter-based mask is keyed by --seed (offset by the rank: shards are not masked identically) unless dropout_seed is given
    config.model_kwargs.setdefault("dropout_seed", config.seed + rank)
    recformer = config.model_type.startswith("RECFORMER")
    if not config.tokenizer_path:
        raise SystemExit("--tokenizer_path <local tokenizer directory> is required (the box is offline)")
    tokenizer = load_tokenizer(config.tokenizer_path)
    reverse = str(config.reverse_sequence).lower() in ("1", "true", "yes")

    def new_model():
        return ModelType[config.model_type].value(
            model_name_or_path=config.model_path, tokenizer_name_or_path=config.tokenizer_path, lora_config=None,
            pooling_method=config.pooling_method, model_kwargs=dict(config.model_kwargs), tokenizer_kwargs=dict(config.tokenizer_kwargs))

    model = new_model()
    pretrain = OrderedDict((k, v.cpu().clone()) for k, v in model.state_dict().items())
    finetune_state_dicts = []
    for path in config.finetune_checkpoint_paths:  # merge_train.py:109-113
        if str(path).startswith("synthetic:"):
            g = torch.Generator().manual_seed(1000 + int(str(path).split(":")[1]))
            sd = OrderedDict((k, v if k.endswith("position_ids") else v + 1e-2 * torch.randn(v.shape, generator=g)) for k, v in pretrain.items())
        else:
            sd = remove_duplicate_prefix(torch.load(path, map_location="cpu"))
        finetune_state_dicts.append(sd)
    n = len(finetune_state_dicts)
    if len(config.data_paths) != n:
        raise SystemExit("--data_paths and --finetune_checkpoint_paths must have the same length")

    # ---- teacher matrices S_d = normalise(seq_d) @ normalise(item_d).T (merge_train.py:114-126), kept in HBM
    score_embeddings = []
    auto = opt["item_embeddings_paths"] == ["auto"]
    if auto and config.train_data_split != "item":
        raise SystemExit("--item_embeddings_paths auto needs --train_data_split item (pseudo users = catalog items)")
    for d in range(n):
        if auto:
            from mergerec_amd.utils import get_data_module

            dm = get_data_module(ModelType[config.model_type], config.batch_size, Path(config.data_paths[d]), config.item_prompt, config.max_attribute_len,
                                 config.max_items, config.max_seq_len, tokenizer, None, config.num_workers, reverse, config.sequence_prompt)
            dm.setup("fit")
            single = new_model()
            single.load_state_dict({k: v for k, v in finetune_state_dicts[d].items() if k != "item_embeddings"})
            probe = RecModule(model=single, evaluator=Evaluator(metrics=config.metric_names, ks=config.ks), negative_sample=None, similarity=config.similarity)
            item_emb = ItemEncoderMixin.encode_items(dm.item_dataloader(), probe)
            seq_emb = item_emb
            del single, probe
        else:
            item_emb = torch.load(opt["item_embeddings_paths"][d], map_location="cpu")
            seq_emb = torch.load(opt["sequence_embeddings_paths"][d], map_location="cpu")
        score_embeddings.append(teacher_scores(seq_emb.to(model.device, torch.float32), item_emb.to(model.device, torch.float32)))

    merged_model = load_merging_module(
        merge_type=MergeType[config.merge_type], learn_type=LearnType[config.learn_type], model=model, pretrain_state_dict=pretrain,
        finetune_state_dicts=[{k: v for k, v in sd.items() if k != "item_embeddings"} for sd in finetune_state_dicts], ignore_keys=set(),
        ties_density=config.ties_density, disable_softmax=not config.use_softmax, initial_per_weight=opt["initial_per_weight"],
        placement="replicated")  # d loss / d alpha contracts the whole gradient with every task vector: data parallel, nothing sliced
    kwargs = ({"coefficient": opt["coefficient"]} if opt["loss_type"].endswith("_KD") else {}) | opt["loss_fn_kwargs"]
    module = DistillSequenceModule(
        merged_model=merged_model, score_embeddings=score_embeddings,
        loss_fn=distill_loss_factory(LossType[opt["loss_type"]], temperature=opt["temperature"], **kwargs), learning_rate=opt["learning_rate"],
        similarity=config.similarity,
        trainable_args_kwargs=({"freeze_global_weight": True, "freeze_global_bias": True} if not config.use_softmax else {}))
    datamodule = (DistillSequenceDataModuleForRecformer if recformer else DistillSequenceDataModule)(
        config.data_paths, tokenizer, config.batch_size, config.max_seq_len, config.max_attribute_len, config.max_items,
        sequence_embeddings=score_embeddings, train_data_split=config.train_data_split, num_workers=config.num_workers,
        valid_ratio=opt["valid_ratio"], reverse_sequence=reverse, num_sequences_per_dataset=opt["num_sequences_per_dataset"],
        sample_method=opt["sample_method"], item_prompt=config.item_prompt, sequence_prompt=config.sequence_prompt)

    class _ItemLoaders:  # the callback reads datamodule.item_dataloaders after setup()
        def __iter__(self):
            return iter(datamodule.item_dataloaders)

        def __len__(self):
            return len(datamodule.item_dataloaders)

    callbacks = [MultiDatasetItemEncodingCallback(_ItemLoaders())]
    save_cb = None
    if rank == 0:  # alpha is identical on every rank (same initial value, averaged gradients): one writer
        save_cb = SaveWeightsCallback(save_dir=opt["weights_dir"], log_every_steps=len(config.data_paths))
        callbacks.insert(0, save_cb)
    weights_checkpoint = None
    if opt["valid_ratio"] is not None:  # merge_train.py:172-175
        weights_checkpoint = WeightCheckpointCallback(monitor=r"val/loss_epoch/dataloader_idx_\d+")
        callbacks.append(weights_checkpoint)
    trainer = DistillTrainer(max_epochs=opt["max_epochs"], max_steps=opt["max_steps"], precision=config.precision, callbacks=callbacks,
                             verbose=rank == 0)
    history = trainer.fit(module, datamodule)
    if weights_checkpoint is not None:
        weights_checkpoint.load_weights(module)  # the best alpha on the held-out pseudo users
    result = dict(history=history, weights=merged_model.serialize_weights(), weights_file=str(save_cb.save_file) if save_cb else None,
                  rank=rank, world_size=world)
    if rank == 0:
        print(f"alpha after {trainer.global_step} steps on {world} rank(s): {result['weights']['per_weights']}")
        print(f"weights written to {save_cb.save_file}")
    if not opt["skip_test"]:  # _test_after_train (merge_train.py:28-67); with several ranks the test itself runs sharded over them
        if rank == 0:
            print("Running test after training...")
        final = new_model()
        final.load_state_dict({k: v.detach() for k, v in merged_model.get_state_dict().items()})
        rec = RecModule(model=final, evaluator=Evaluator(metrics=config.metric_names, ks=config.ks), negative_sample=None, similarity=config.similarity)
        metric_dict, metrics, _, _ = test_model(
            rec, ModelType[config.model_type], [Path(p) for p in config.test_data_paths], tokenizer, config.batch_size, config.max_seq_len,
            config.max_attribute_len, config.max_items, config.num_workers, config.sequence_prompt, config.item_prompt, reverse, config.precision,
            config.test_data_split, metrics_path=config.metrics_path, predictions_path=config.predictions_path)
        if rank == 0:
            print(f"Test metrics after training: {metric_dict}")
        result["test_metrics"] = metric_dict
    if opt["result_path"]:
        torch.save(result, f"{opt['result_path']}.rank{rank}.pt")
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
