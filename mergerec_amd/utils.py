"""Mirror of the reference's top-level utils.py (remove_duplicate_prefix :17-29, test_model :32-134,
save_predictions :178-214) plus the minimal Trainer the test loop needs when lightning is absent."""
from __future__ import annotations

import ast
import csv
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence

import torch

from .module.callbacks import ItemEncodingCallback
from .module.recommender import RecModule
from .parallel import allreduce_mean_grads, shard_indices


def remove_duplicate_prefix(state_dict):
    new = {}
    for k, v in state_dict.items():
        if k.startswith("model."):
            new[k.replace("model.", "", 1)] = v
        else:
            print(f"Keeping key {k} without model. prefix")
            new[k] = v
    return new


def load_alpha_file(path: Path, line: int) -> dict:
    """merge_test.py:67-68 parses each line with eval(); literal_eval accepts the same dict reprs safely."""
    rows = [ast.literal_eval(l) for l in Path(path).read_text().strip().splitlines()]
    return rows[line]["weights"]


def precision_to_gemm_mode(precision) -> Optional[str]:
    """Lightning precision flag -> encoder arithmetic.  "32-true" keeps the model's mode (library default bf16x6: fp32-grade).  The
    reference's default "bf16-mixed" (torch autocast: 8-bit-mantissa products) maps to "bf16x3", the fastest arithmetic built -- three
    bf16 MFMA products per fp32 product, ~1e-6 on the embeddings, i.e. orders of magnitude tighter than what that flag asks for."""
    p = str(precision)
    if p in ("32-true", "32"):
        return None
    if p in ("bf16-mixed", "bf16", "16-mixed", "16", "bf16-true", "16-true"):
        print(f"precision={p!r}: running the bf16x3 split-precision kernels (fp32 accumulation; stricter than autocast {p})")
        return "bf16x3"
    if p in ("64-true", "64"):
        raise NotImplementedError("fp64 is not built")
    raise ValueError(f"unknown precision {precision!r}")


class Trainer:
    """``lightning.Trainer(...).test(module, dataloader)`` hook order: callbacks' on_test_epoch_start,
    module.on_test_epoch_start, test_step per batch (moved to the module's device), on_test_epoch_end."""

    def __init__(self, precision: str = "32-true", callbacks: Sequence = (), coalesce_tokens: int = 65536, **_):
        self.gemm_mode = precision_to_gemm_mode(precision)
        self.callbacks = list(callbacks)
        self.coalesce_tokens = coalesce_tokens  # 0: one kernel pass per dataloader batch, like the reference

    @torch.no_grad()
    def test(self, module: RecModule, dataloader: Iterable, verbose: bool = False) -> List[Dict[str, float]]:
        module.trainer = self
        module.eval()
        if self.gemm_mode is not None and hasattr(module.model, "set_gemm_mode"):
            module.model.set_gemm_mode(self.gemm_mode)
        for cb in self.callbacks:
            if hasattr(cb, "on_test_epoch_start"):
                cb.on_test_epoch_start(self, module)
        module.on_test_epoch_start()
        from .data import coalesce_batches

        stream = coalesce_batches(dataloader, self.coalesce_tokens) if self.coalesce_tokens else dataloader
        for i, batch in enumerate(stream):
            module.test_step(batch.to(module.device), i)
        metrics = module.on_test_epoch_end()
        return [dict(metrics)]


def test_model(module: RecModule, item_dataloaders: Sequence[Iterable], sequence_dataloaders: Sequence[Iterable],
               data_names: Sequence[str], precision: str = "32-true", metrics_path: Optional[Path] = None,
               predictions_path: Optional[Path] = None, item_embeddings_path: Optional[Path] = None,
               user_embeddings_path: Optional[Path] = None):
    """utils.py:32-134 from the point where the per-domain dataloaders exist (the datamodule / tokeniser
    side is a 'next' row).  Returns (metric_dict, metrics, scores, labels) like the reference; ``scores``
    entries are None unless predictions are being saved (the fused path does not materialise them)."""
    cb = ItemEncodingCallback()
    trainer = Trainer(precision=precision, callbacks=[cb])
    module.keep_scores = predictions_path is not None
    metric_dict, metrics, scores, labels, item_embs, user_embs = {}, [], [], [], [], []
    for i, (item_dl, seq_dl) in enumerate(zip(item_dataloaders, sequence_dataloaders)):
        cb.item_dataloader = item_dl
        module.item_embeddings = None  # utils.py:110: catalog re-encoded per domain
        metric = trainer.test(module, seq_dl, verbose=False)
        scores.append(None if module.eval_scores is None else module.eval_scores.detach().cpu().clone())
        labels.append(module.eval_labels.detach().cpu().clone())
        item_embs.append(module.item_embeddings.detach().cpu().clone())
        user_embs.append(module.eval_user_embeddings.detach().cpu().clone())
        metrics.append(metric[0])
        metric_dict.update({f"test/dataset_{i}/{k}": v for k, v in metric[0].items()})
    save_predictions(data_names, item_embs, item_embeddings_path, labels, metrics, metrics_path, predictions_path, scores,
                     user_embs, user_embeddings_path)
    return metric_dict, metrics, scores, labels


class DistillTrainer:
    """The slice of ``lightning.Trainer.fit`` that merge_train.py uses (merge_train.py:178-196): epochs over the training
    dataloader until max_steps / max_epochs, optimizer from ``configure_optimizers``, callback hooks on_train_epoch_start /
    on_train_batch_end / on_train_epoch_end / teardown, optional validation dataloaders after every epoch."""

    def __init__(self, max_epochs: Optional[int] = None, max_steps: Optional[int] = None, callbacks: Sequence = (), precision: str = "32-true",
                 coalesce_tokens: int = 65536, log_every_n_steps: int = 1, verbose: bool = True):
        if precision_to_gemm_mode(precision) is not None:
            raise NotImplementedError("the optimisation loop runs the exact-fp32 training graph; use precision 32-true")
        if max_epochs is None and (max_steps is None or max_steps < 0):
            raise ValueError("max_steps or max_epochs is required")
        self.max_epochs, self.max_steps, self.callbacks = max_epochs, max_steps, list(callbacks)
        self.coalesce_tokens, self.log_every_n_steps, self.verbose = coalesce_tokens, log_every_n_steps, verbose
        self.current_epoch = 0
        self.global_step = 0
        self.history: List[float] = []
        self.callback_metrics: Dict[str, torch.Tensor] = {}

    def _train_loader(self, datamodule):
        """One rank: the datamodule's own shuffled loader.  Several ranks (torch.distributed initialised): the same dataset and
        collator, the epoch's permutation dealt round-robin over the ranks."""
        import torch.distributed as dist
        from torch.utils.data import DataLoader, Subset

        loader = datamodule.train_dataloader()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return loader
        idx = shard_indices(len(loader.dataset), dist.get_rank(), dist.get_world_size(), 1234 + self.current_epoch)
        return DataLoader(Subset(loader.dataset, idx), batch_size=loader.batch_size, collate_fn=loader.collate_fn, shuffle=False,
                          num_workers=loader.num_workers)

    def _hook(self, name, *args):
        for cb in self.callbacks:
            if hasattr(cb, name):
                getattr(cb, name)(self, *args)

    def fit(self, module, datamodule):
        module.trainer = self
        datamodule.setup("fit")
        opt = module.configure_optimizers()
        trainable = [p for grp in opt.param_groups for p in grp["params"]]
        done = False
        while not done:
            module.train()
            self._hook("on_train_epoch_start", module)
            for batch_idx, batch in enumerate(self._train_loader(datamodule)):  # reload_dataloaders_every_n_epochs=1
                opt.zero_grad(set_to_none=True)
                loss = module.training_step(batch.to(module.device), batch_idx)
                loss.backward()
                allreduce_mean_grads(trainable)  # data parallel: the only exchange of the step (no-op on one rank)
                opt.step()
                self.global_step += 1
                self.history.append(float(loss.detach()))
                if self.verbose and self.global_step % self.log_every_n_steps == 0:
                    print(f"step {self.global_step}: train/loss {self.history[-1]:.6f}")
                self._hook("on_train_batch_end", module, loss, batch, batch_idx)
                if self.max_steps is not None and self.max_steps >= 0 and self.global_step >= self.max_steps:
                    done = True
                    break
            self._hook("on_train_epoch_end", module)
            vals = datamodule.val_dataloader() if hasattr(datamodule, "val_dataloader") else []
            if vals:
                module.eval()
                module.on_validation_epoch_start()
                for di, dl in enumerate(vals):
                    tot, cnt = 0.0, 0
                    for bi, batch in enumerate(dl):
                        n = len(batch.dataset_indexes)
                        tot += float(module.validation_step(batch.to(module.device), bi, di)) * n
                        cnt += n
                    # Lightning's on_epoch aggregation of self.log("val/loss", ...): batch-size-weighted mean per dataloader
                    self.callback_metrics[f"val/loss_epoch/dataloader_idx_{di}"] = torch.tensor(tot / max(cnt, 1))
                module.on_validation_epoch_end()
                self._hook("on_validation_epoch_end", module)
            self.current_epoch += 1
            if self.max_epochs is not None and self.current_epoch >= self.max_epochs:
                done = True
        self._hook("teardown", module, "fit")
        return self.history


def get_data_module(model_type, batch_size, data_path, item_prompt, max_attribute_len, max_items, max_seq_len, model_tokenizer,
                    negative_sample_config, num_workers, reverse_sequence, sequence_prompt):
    """utils.py:137-175 (_get_data_module): the Recformer datamodule for the RECFORMER* model types, the text one otherwise."""
    from .datamodule import RecDataModule, RecDataModuleForRecformer

    name = getattr(model_type, "name", str(model_type)).upper()
    if name.startswith("RECFORMER"):
        return RecDataModuleForRecformer(dataset_path=data_path, tokenizer=model_tokenizer, batch_size=batch_size, max_seq_len=max_seq_len,
                                         max_attribute_len=max_attribute_len, max_items=max_items, num_workers=num_workers,
                                         negative_sample=negative_sample_config)
    return RecDataModule(dataset_path=data_path, tokenizer=model_tokenizer, batch_size=batch_size, max_seq_len=max_seq_len,
                         max_attribute_len=max_attribute_len, max_items=max_items, num_workers=num_workers,
                         negative_sample=negative_sample_config, sequence_prompt=sequence_prompt, item_prompt=item_prompt,
                         reverse_sequence=reverse_sequence)


def test_model_from_paths(module: RecModule, model_type, data_paths: Sequence[Path], model_tokenizer, batch_size: int, max_seq_len: int,
                          max_attribute_len: int, max_items: Optional[int], num_workers: int, sequence_prompt: Optional[str],
                          item_prompt: Optional[str], reverse_sequence: bool, precision: str, data_split: str,
                          metrics_path: Optional[Path] = None, predictions_path: Optional[Path] = None,
                          item_embeddings_path: Optional[Path] = None, user_embeddings_path: Optional[Path] = None):
    """The reference's ``test_model`` signature (utils.py:30-134): dataset directories in, metrics out."""
    item_dls, seq_dls = [], []
    for data_path in data_paths:
        dm = get_data_module(model_type, batch_size, Path(data_path), item_prompt, max_attribute_len, max_items, max_seq_len, model_tokenizer,
                             None, num_workers, reverse_sequence, sequence_prompt)
        dm.setup("fit")
        item_dls.append(dm.item_dataloader())
        if data_split == "val":
            seq_dls.append(dm.val_dataloader())
        elif data_split == "test":
            seq_dls.append(dm.test_dataloader())
        else:
            raise ValueError(f"Unknown data split: {data_split}")
    return test_model(module, item_dls, seq_dls, [Path(p).name for p in data_paths], precision=precision, metrics_path=metrics_path,
                      predictions_path=predictions_path, item_embeddings_path=item_embeddings_path, user_embeddings_path=user_embeddings_path)


def save_predictions(data_names, item_embeddings, item_embeddings_path, labels, metrics, metrics_path, predictions_path,
                     scores, user_embeddings, user_embeddings_path):
    if metrics_path is not None:  # utils.py:191-196: CSV indexed by dataset dir name
        cols = list(metrics[0].keys()) if metrics else []
        with open(metrics_path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["dataset"] + cols)
            for name, m in zip(data_names, metrics):
                w.writerow([name] + [m[c] for c in cols])
        print(f"Saved metrics to {metrics_path}")
    if predictions_path is not None:
        torch.save({n: {"scores": s, "labels": l} for n, s, l in zip(data_names, scores, labels)}, predictions_path)
        print(f"Saved predictions to {predictions_path}")
    if item_embeddings_path is not None:
        torch.save(item_embeddings, item_embeddings_path)
        print(f"Saved item embeddings to {item_embeddings_path}")
    if user_embeddings_path is not None:
        torch.save(user_embeddings, user_embeddings_path)
        print(f"Saved user embeddings to {user_embeddings_path}")
