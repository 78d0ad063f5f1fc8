"""Mirror of the reference's top-level utils.py (remove_duplicate_prefix :17-29, test_model :32-134 with the reference's argument
list, save_predictions :178-214) plus the minimal Trainer the test loop needs when lightning is absent."""
from __future__ import annotations

import ast
import csv
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Sequence

import torch

from .engine import check_module_inputs
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
    """Lightning precision flag -> encoder arithmetic.  "32-true" keeps the model's mode (library default bf16x6: fp32-grade, no range
    limits).  The reference's default "bf16-mixed" (torch autocast: 8-bit-mantissa products) maps to "f16x3" -- three fp16 MFMA products
    per fp32 product, ~2^-21 each: at the reference's own fp32 rounding level on trained-like weights (fixture g22), orders of magnitude
    tighter than what that flag asks for, at the cost of the former choice "bf16x3" (2^-16 per product: 10x the fp32 noise on g22;
    still selectable through ``gemm_mode`` / MERGEREC_GEMM_MODE)."""
    p = str(precision)
    if p in ("32-true", "32"):
        return None
    if p in ("bf16-mixed", "bf16", "16-mixed", "16", "bf16-true", "16-true"):
        print(f"precision={p!r}: running the f16x3 split-precision kernels (fp32 accumulation; stricter than autocast {p})")
        return "f16x3"
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
        """One evaluation epoch.  With ``torch.distributed`` initialised on several ranks (one per GPU) every rank calls this with the
        SAME dataloader: the catalog rows (callback) and the user sequences are dealt over the ranks (``parallel.ShardedLoader``,
        token-balanced), each rank scores its users against the full item matrix, and the per-user results are gathered at the
        epoch end -- every rank returns the single-process metrics."""
        from .data import coalesce_batches
        from .parallel import ShardedLoader

        module.trainer = self
        module.eval()
        if self.gemm_mode is not None and hasattr(module.model, "set_gemm_mode"):
            module.model.set_gemm_mode(self.gemm_mode)
        for cb in self.callbacks:
            if hasattr(cb, "on_test_epoch_start"):
                cb.on_test_epoch_start(self, module)
        module.on_test_epoch_start()
        shard = ShardedLoader(dataloader, balance=True)
        module._shard = shard
        stream = coalesce_batches(shard, self.coalesce_tokens) if self.coalesce_tokens else shard
        for i, batch in enumerate(stream):
            module.test_step(batch.to(module.device), i)
        metrics = module.on_test_epoch_end()
        module._shard = None
        return [dict(metrics)]


def _domain_loaders(model_type, data_path, model_tokenizer, batch_size, max_seq_len, max_attribute_len, max_items, num_workers,
                    sequence_prompt, item_prompt, reverse_sequence, data_split, vocab):
    """(name, item dataloader, sequence dataloader) of one ``data_paths`` entry: a dataset directory in the reference's JSON format
    (utils.py:62-81: datamodule + tokenizer), or -- this build's additions -- a directory / file with pre-tokenised tensors
    (``tokenized.pt``) or a ``synthetic:<Name>[:M[:U]]`` spec (mergerec_amd/data.py)."""
    if data_split not in ("val", "test"):
        raise ValueError(f"Unknown data split: {data_split}")
    spec = str(data_path)
    if not spec.startswith("synthetic:") and (Path(spec) / "train.json").exists():
        if model_tokenizer is None:
            raise ValueError("a tokenizer is required for dataset directories in the JSON format (pass tokenizer_name_or_path = a local "
                             "tokenizer directory to the model; the box is offline)")
        dm = get_data_module(model_type, batch_size, Path(spec), item_prompt, max_attribute_len, max_items, max_seq_len, model_tokenizer,
                             None, num_workers, reverse_sequence, sequence_prompt)
        dm.setup("fit")
        return Path(spec).name, dm.item_dataloader(), (dm.val_dataloader() if data_split == "val" else dm.test_dataloader())
    from .data import load_domain

    kind = "recformer" if getattr(model_type, "name", str(model_type)).upper().startswith("RECFORMER") else "roberta"
    dom = load_domain(spec, kind=kind, vocab=vocab)
    return dom.name, dom.item_dataloader(batch_size), dom.sequence_dataloader(batch_size)


def test_model(module: RecModule, model_type, data_paths: Sequence, model_tokenizer, batch_size: int, max_seq_len: int,
               max_attribute_len: int, max_items: Optional[int], num_workers: int, sequence_prompt: Optional[str],
               item_prompt: Optional[str], reverse_sequence: bool, precision: str, data_split: str,
               metrics_path: Optional[Path] = None, predictions_path: Optional[Path] = None,
               item_embeddings_path: Optional[Path] = None, user_embeddings_path: Optional[Path] = None) -> tuple:
    """The reference's ``test_model`` (utils.py:32-134), argument for argument: dataset paths in, (metric_dict, metrics, scores,
    labels) out; ``metric_dict`` keys ``test/dataset_{i}/{k}``.  ``scores[i]`` is domain i's (users, items) matrix as upstream: written by the
    scoring kernel when ``predictions_path`` is given, otherwise produced on first access (``LazyScores``: same bits, no cost if unread).  Runs sharded over the ranks when launched under
    ``torch.distributed.run`` (see ``Trainer.test``)."""
    vocab = getattr(getattr(module.model, "spec", None), "vocab", 50265)
    names, item_dls, seq_dls = [], [], []
    for data_path in data_paths:
        name, item_dl, seq_dl = _domain_loaders(model_type, data_path, model_tokenizer, batch_size, max_seq_len, max_attribute_len, max_items,
                                                num_workers, sequence_prompt, item_prompt, reverse_sequence, data_split, vocab)
        names.append(name)
        item_dls.append(item_dl)
        seq_dls.append(seq_dl)
    return test_model_on_dataloaders(module, item_dls, seq_dls, names, precision=precision, metrics_path=metrics_path,
                                     predictions_path=predictions_path, item_embeddings_path=item_embeddings_path,
                                     user_embeddings_path=user_embeddings_path)


def test_model_on_dataloaders(module: RecModule, item_dataloaders: Sequence[Iterable], sequence_dataloaders: Sequence[Iterable],
                              data_names: Sequence[str], precision: str = "32-true", metrics_path: Optional[Path] = None,
                              predictions_path: Optional[Path] = None, item_embeddings_path: Optional[Path] = None,
                              user_embeddings_path: Optional[Path] = None):
    """utils.py:83-134: the per-domain test loop of ``test_model`` from the point where the dataloaders exist (callers that already
    hold dataloaders -- tests, synthetic domains -- enter here)."""
    cb = ItemEncodingCallback()
    trainer = Trainer(precision=precision, callbacks=[cb])
    module.keep_scores = predictions_path is not None
    metric_dict, metrics, scores, labels, item_embs, user_embs = {}, [], [], [], [], []
    for i, (item_dl, seq_dl) in enumerate(zip(item_dataloaders, sequence_dataloaders)):
        cb.item_dataloader = item_dl
        module.item_embeddings = None  # utils.py:110: catalog re-encoded per domain
        metric = trainer.test(module, seq_dl, verbose=False)
        scores.append(module.eval_scores.detach().cpu() if module.keep_scores and module.eval_scores is not None else None)
        labels.append(module.eval_labels.detach().cpu().clone())
        item_embs.append(module.item_embeddings.detach().cpu().clone())
        user_embs.append(module.eval_user_embeddings.detach().cpu().clone())
        metrics.append(metric[0])
        metric_dict.update({f"test/dataset_{i}/{k}": v for k, v in metric[0].items()})
    save_predictions(data_names, item_embs, item_embeddings_path, labels, metrics, metrics_path, predictions_path, scores,
                     user_embs, user_embeddings_path)
    if predictions_path is None:  # the reference returns every domain's (users, items) matrix (utils.py:113): produced on first access
        scores = LazyScores(user_embs, item_embs, module.device)
    return metric_dict, metrics, scores, labels


class LazyScores(Sequence):
    """``test_model``'s third return value when no predictions file was asked for: a list-like over the domains whose element i is the
    (users_i, items_i) fp32 score matrix, computed on first access by the scoring path's exact-fp32 product from the embeddings the
    evaluation kept (the same bits the kernel ranked), then cached.  The reference materialises these matrices on the host for every
    call; here a caller that never looks pays nothing."""

    def __init__(self, user_embs, item_embs, device):
        self._u, self._e, self._dev = list(user_embs), list(item_embs), device
        self._cache = [None] * len(self._u)

    def __len__(self):
        return len(self._u)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if self._cache[i] is None:
            from .module.recommender import score_matrix

            self._cache[i] = score_matrix(self._u[i], self._e[i], self._dev)
        return self._cache[i]


class DistillTrainer:
    """The slice of ``lightning.Trainer.fit`` that merge_train.py uses (merge_train.py:178-196): epochs over the training
    dataloader until max_steps / max_epochs, optimizer from ``configure_optimizers``, callback hooks on_train_epoch_start /
    on_train_batch_end / on_train_epoch_end / teardown, optional validation dataloaders after every epoch."""

    def __init__(self, max_epochs: Optional[int] = None, max_steps: Optional[int] = None, callbacks: Sequence = (), precision: str = "32-true",
                 coalesce_tokens: int = 65536, log_every_n_steps: int = 1, verbose: bool = True):
        # Every Lightning precision string is accepted.  "32-true": exact-fp32 products.  The reduced-precision flags (the reference's default
        # bf16-mixed, 16-mixed, ...): by the batch's token count -- the recipe's step is 16 short pseudo-user sequences (~600 tokens) against
        # freshly merged weights, where re-splitting the weights for the bf16x3 graph costs more than its faster products save (8.4 vs 11.5 ms
        # at BLaIR-base x 8), so it stays on the exact-fp32 tile kernel; from ~1,100 tokens per step on the split graph wins (4,782 tokens:
        # 20.5 vs 39.9 ms) and is taken (``TaskVectorMergingModuleBase.train_mode = "auto"``)
        self.train_mode = "f32" if precision_to_gemm_mode(precision) is None else "auto"
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
        if hasattr(module, "merged_model"):
            module.merged_model.train_mode = self.train_mode
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
                self.history.append(loss.detach())  # stays on the device: no host sync per step unless this step is logged
                if self.global_step == 1 or (self.verbose and self.global_step % self.log_every_n_steps == 0):
                    check_module_inputs(module)  # bad ids must not train alpha on clamped embeddings: first step + every logged step
                if self.verbose and self.global_step % self.log_every_n_steps == 0:
                    print(f"step {self.global_step}: train/loss {float(self.history[-1]):.6f}")
                self._hook("on_train_batch_end", module, loss, batch, batch_idx)
                if self.max_steps is not None and self.max_steps >= 0 and self.global_step >= self.max_steps:
                    done = True
                    break
            check_module_inputs(module)
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
                check_module_inputs(module)
                module.on_validation_epoch_end()
                self._hook("on_validation_epoch_end", module)
            self.current_epoch += 1
            if self.max_epochs is not None and self.current_epoch >= self.max_epochs:
                done = True
        self._hook("teardown", module, "fit")
        self.history = [float(x) for x in self.history]
        return self.history


class FinetuneTrainer:
    """``lightning.Trainer(...).fit(module, datamodule)`` followed by ``.test(module, datamodule, ckpt_path="best")`` as
    finetune_train.py:78-114 configures them: gradient accumulation, global-norm clipping, per-step LR schedule, a validation pass
    after every epoch, ModelCheckpoint(monitor, mode="max", save_top_k=1, "epoch_{epoch:02d}") and EarlyStopping(monitor, patience).

    One step = ONE packed encoder forward + backward over [sequences; targets (; negatives)] through the HIP training graph, the
    gradient accumulated in one arena, and one fused clip + AdamW launch per optimizer step.  With torch.distributed initialised the
    epoch's permutation is dealt over the ranks and the gradient arena is averaged in ONE all-reduce per optimizer step."""

    def __init__(self, max_epochs: int, accumulate_grad_batches: int = 1, gradient_clip_val: Optional[float] = None, precision: str = "32-true",
                 callbacks: Sequence = (), monitor: str = "val/NDCG@10", patience: int = 5, default_root_dir="MergeRecFineTune",
                 log_every_n_steps: int = 50, coalesce_tokens: int = 65536, max_steps: Optional[int] = None, verbose: bool = True):
        # "32-true": exact-fp32 training graph, the model's own evaluation arithmetic; the reference's default "bf16-mixed" (autocast:
        # 8-bit-mantissa products): bf16x3 split-precision products (fp32 accumulation, ~1e-6 relative) for training and evaluation
        self.gemm_mode = precision_to_gemm_mode(precision)
        self.train_mode = "f32" if self.gemm_mode is None else "bf16x3"
        self.max_epochs, self.max_steps = int(max_epochs), max_steps
        self.accumulate_grad_batches = max(1, int(accumulate_grad_batches))
        self.gradient_clip_val = gradient_clip_val
        self.callbacks = list(callbacks)
        self.monitor, self.patience = monitor, patience
        self.root = Path(default_root_dir)
        self.log_every_n_steps, self.coalesce_tokens, self.verbose = log_every_n_steps, coalesce_tokens, verbose
        self.current_epoch = self.global_step = 0
        self.estimated_stepping_batches = 0
        self.callback_metrics: Dict[str, float] = {}
        self.history: List[float] = []
        self.lr_history: List[float] = []
        self.best_score: Optional[float] = None
        self.best_model_path: Optional[Path] = None
        self.optimizer = None

    def _hook(self, name, module):
        for cb in self.callbacks:
            if hasattr(cb, name):
                getattr(cb, name)(self, module)

    @staticmethod
    def _dist():
        import torch.distributed as dist

        on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        return (dist.get_rank(), dist.get_world_size()) if on else (0, 1)

    def _train_loader(self, datamodule):
        from torch.utils.data import DataLoader, Subset

        loader = datamodule.train_dataloader()
        rank, world = self._dist()
        if world == 1:
            return loader
        idx = shard_indices(len(loader.dataset), rank, world, 1234 + self.current_epoch)
        return DataLoader(Subset(loader.dataset, idx), batch_size=loader.batch_size, collate_fn=loader.collate_fn, shuffle=False,
                          num_workers=loader.num_workers, drop_last=True)

    # -- validation / checkpoint ---------------------------------------------------------------------
    @torch.no_grad()
    def _validate(self, module, datamodule) -> Dict[str, float]:
        from .data import coalesce_batches

        module.eval()
        if self.gemm_mode is not None and hasattr(module.model, "set_gemm_mode"):
            module.model.set_gemm_mode(self.gemm_mode)
        self._hook("on_validation_epoch_start", module)
        module.on_validation_epoch_start()
        dl = datamodule.val_dataloader()
        stream = coalesce_batches(dl, self.coalesce_tokens) if self.coalesce_tokens else dl
        for i, batch in enumerate(stream):
            module.validation_step(batch.to(module.device), i)
        metrics = module.on_validation_epoch_end()
        self.callback_metrics.update(metrics)
        return metrics

    def _save_checkpoint(self, module) -> Path:
        """Lightning's checkpoint layout as far as scripts/extract.py reads it: ``state_dict`` with the module's parameter names
        (``model.model.*`` + ``item_embeddings``)."""
        d = self.root / "checkpoints"
        d.mkdir(parents=True, exist_ok=True)
        path = d / f"epoch_{self.current_epoch:02d}.ckpt"
        sd = {"model." + k: v.detach().cpu().clone() for k, v in module.model.state_dict().items()}
        if module.item_embeddings is not None:
            sd["item_embeddings"] = module.item_embeddings.detach().cpu().clone()
        torch.save({"state_dict": sd, "epoch": self.current_epoch, "global_step": self.global_step,
                    "monitor": self.monitor, "score": self.best_score}, path)
        return path

    # -- fit -----------------------------------------------------------------------------------------
    def fit(self, module, datamodule):
        module.trainer = self
        if getattr(datamodule, "train_dataset", None) is None:
            datamodule.setup("fit")
        rank, world = self._dist()
        acc = self.accumulate_grad_batches
        n_batches = len(self._train_loader(datamodule))
        self.estimated_stepping_batches = -(-n_batches // acc) * max(self.max_epochs, 1)
        if self.max_steps is not None and self.max_steps >= 0:
            self.estimated_stepping_batches = min(self.estimated_stepping_batches, self.max_steps)
        opt = self.optimizer = module.configure_optimizers()
        leaf = module.model.train_leaf()
        if self.train_mode == "bf16x3" and module.model.spec.hidden % 128 == 0:
            module.model.train_mode = "bf16x3"
        wait, stop = 0, False
        while not stop and self.current_epoch < self.max_epochs:
            module.train()
            self._hook("on_train_epoch_start", module)
            loader = self._train_loader(datamodule)
            last = len(loader) - 1
            leaf.grad = None
            for batch_idx, batch in enumerate(loader):
                loss = module.training_step(batch.to(module.device), batch_idx)
                (loss / acc).backward()  # accumulates into the one gradient arena
                self.history.append(loss.detach())
                if (batch_idx + 1) % acc and batch_idx != last:
                    continue
                allreduce_mean_grads([leaf])
                self.lr_history.append(opt.step(leaf.grad))
                leaf.grad = None
                module.model.arena_changed()
                self.global_step += 1
                if self.global_step == 1 or (self.verbose and self.global_step % self.log_every_n_steps == 0):
                    check_module_inputs(module)  # first optimizer step + every logged step (they synchronise anyway)
                if self.verbose and self.global_step % self.log_every_n_steps == 0:
                    print(f"epoch {self.current_epoch} step {self.global_step}: train/loss {float(self.history[-1]):.6f} lr {self.lr_history[-1]:.3e}",
                          flush=True)
                if self.max_steps is not None and 0 <= self.max_steps <= self.global_step:
                    stop = True
                    break
            module.model.weights_updated()  # the arena changed under the inference path's bf16 weight pieces
            check_module_inputs(module)
            self._hook("on_train_epoch_end", module)
            metrics = self._validate(module, datamodule)
            score = metrics.get(self.monitor)
            if score is None:
                raise KeyError(f"monitored metric {self.monitor!r} not produced by validation: {sorted(metrics)}")
            if self.verbose:
                print(f"epoch {self.current_epoch}: " + ", ".join(f"{k} {v:.5f}" for k, v in sorted(metrics.items())), flush=True)
            if self.best_score is None or score > self.best_score:  # ModelCheckpoint(mode="max", save_top_k=1) + EarlyStopping
                self.best_score, wait = score, 0
                if rank == 0:
                    old = self.best_model_path
                    self.best_model_path = self._save_checkpoint(module)
                    if old is not None and old != self.best_model_path and old.exists():
                        old.unlink()
            else:
                wait += 1
                if wait >= self.patience:
                    if self.verbose:
                        print(f"early stopping: {self.monitor} did not improve for {wait} validation epochs (best {self.best_score:.5f})")
                    stop = True
            self.current_epoch += 1
        self.history = [float(x) for x in self.history]
        return self.history

    # -- test ----------------------------------------------------------------------------------------
    def test(self, module, datamodule, ckpt_path: Optional[str] = "best"):
        if ckpt_path == "best":
            ckpt_path = self.best_model_path
        if ckpt_path is not None:
            sd = dict(torch.load(ckpt_path, map_location="cpu")["state_dict"])
            items = sd.pop("item_embeddings", None)
            module.model.load_state_dict(remove_duplicate_prefix(sd))
            module.item_embeddings = None if items is None else torch.nn.Parameter(items.to(module.device), requires_grad=False)
        tester = Trainer(precision="32-true", callbacks=self.callbacks, coalesce_tokens=self.coalesce_tokens)
        tester.gemm_mode = self.gemm_mode
        return tester.test(module, datamodule.test_dataloader())


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


def save_predictions(data_names, item_embeddings, item_embeddings_path, labels, metrics, metrics_path, predictions_path,
                     scores, user_embeddings, user_embeddings_path):
    """utils.py:178-214.  ``data_names``: dataset directory names (or Paths, whose ``.name`` is used).  One writer: rank 0."""
    from .parallel import world

    if world()[0] != 0:
        return
    data_names = [getattr(n, "name", n) if isinstance(n, Path) else n for n in data_names]
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
