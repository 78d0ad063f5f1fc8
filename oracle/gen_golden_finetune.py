#!/usr/bin/env python3
"""Generate tests/golden/g9_finetune.pt from the reference's RecModule (BUILD CONTAINER ONLY).

TEST INFRASTRUCTURE, same rules as gen_golden.py: reads /root/reference at generation time, commits tensors only.
``rec_retrieval/module/recommender/module.py`` is loaded standalone by path.  Its imports that cannot be satisfied here are
replaced by empty stand-ins that carry no arithmetic: ``lightning`` (absent; ``LightningModule`` -> a bare nn.Module with no-op
``log``), ``..models`` (needs peft; only the ``BaseModel`` annotation is used).  ``...configs`` / ``...types`` / ``...evaluator`` are
the reference's own modules through the importer of gen_golden.py.  What is recorded:

  scores   RecModule._forward_negative_sample for IN_BATCH / SAMPLE / IN_BATCH_SAMPLE and ``training_step`` on a toy encoder
           (embedding-bag + Linear + LayerNorm: parameter names with "bias" and "LayerNorm.weight", so both optimizer groups exist)
  optim    RecModule.configure_optimizers() (AdamW groups + linear warm-up schedule) driven for several steps the way Lightning
           drives it -- backward, clip_grad_norm_(gradient_clip_val), optimizer.step(), scheduler.step() -- recording, per step,
           the learning rate, the loss, every gradient and every parameter after the step.
"""
from __future__ import annotations

import sys
import types
from collections import OrderedDict
from pathlib import Path

import torch
from torch import nn

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.gen_golden import OUT, REF, install_reference_importer, load_by_path  # noqa: E402


class ToyEncoder(nn.Module):
    """BatchEncoding -> (B, d): masked mean of token embeddings -> Linear -> LayerNorm.  Stands where BLaIR stands."""

    def __init__(self, vocab=50, d=16, seed=5):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.tokenizer = None
        self.word = nn.Embedding(vocab, d)
        self.dense = nn.Linear(d, d)
        self.LayerNorm = nn.LayerNorm(d)
        with torch.no_grad():
            self.word.weight.copy_(torch.randn(vocab, d, generator=g))
            self.dense.weight.copy_(torch.randn(d, d, generator=g) * 0.3)
            self.dense.bias.copy_(torch.randn(d, generator=g) * 0.1)
            self.LayerNorm.weight.copy_(1 + 0.1 * torch.randn(d, generator=g))
            self.LayerNorm.bias.copy_(0.1 * torch.randn(d, generator=g))

    def forward(self, batch):
        m = batch["attention_mask"].unsqueeze(-1).float()
        x = (self.word(batch["input_ids"]) * m).sum(1) / m.sum(1)
        return self.LayerNorm(self.dense(x))


def toy_batch(B, L, vocab, g):
    lens = torch.randint(2, L + 1, (B,), generator=g)
    ids = torch.randint(0, vocab, (B, L), generator=g)
    mask = (torch.arange(L).view(1, L) < lens.view(B, 1)).long()
    return {"input_ids": ids, "attention_mask": mask}


def main():
    torch.set_num_threads(4)
    install_reference_importer()
    # stand-ins without arithmetic (see the header)
    lightning = types.ModuleType("lightning")

    class LightningModule(nn.Module):
        trainer = None

        def log(self, *a, **k):
            pass

        def log_dict(self, *a, **k):
            pass

    lightning.LightningModule = LightningModule
    sys.modules["lightning"] = lightning
    for name in ("rec_retrieval.module", "rec_retrieval.module.recommender"):
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
    models = types.ModuleType("rec_retrieval.module.models")
    models.BaseModel = nn.Module
    sys.modules["rec_retrieval.module.models"] = models

    from transformers import BatchEncoding

    from rec_retrieval.configs import NegativeSampleConfig
    from rec_retrieval.evaluator import Evaluator
    from rec_retrieval.types import BatchSequenceWithNegative

    mod = load_by_path("rec_retrieval.module.recommender.module", REF / "rec_retrieval/module/recommender/module.py")

    g = torch.Generator().manual_seed(2024)
    vocab, d, B, k = 50, 16, 6, 3
    enc = lambda b: BatchEncoding(b)
    seq, tgt, neg = toy_batch(B, 9, vocab, g), toy_batch(B, 5, vocab, g), toy_batch(B * k, 5, vocab, g)

    def make(ns, **kw):
        return mod.RecModule(model=ToyEncoder(vocab, d), evaluator=Evaluator(metrics=["NDCG"], ks=[1]), negative_sample=ns, similarity="cosine",
                             **kw)

    # ---------------------------------------------------------------- scores / labels / loss per negative-sampling mode
    score_cases = []
    for ns in (NegativeSampleConfig(in_batch=True), NegativeSampleConfig(k=k), NegativeSampleConfig(k=k, in_batch=True)):
        m = make(ns, temperature=0.05)
        with torch.no_grad():
            reps = [torch.nn.functional.normalize(m.model(b), dim=-1) for b in (seq, tgt, neg)]
            scores, labels = m.forward(BatchSequenceWithNegative(sequence=enc(seq), target=enc(tgt),
                                                                 negatives=None if ns.k is None else enc(neg)))
            loss = m.training_step(BatchSequenceWithNegative(sequence=enc(seq), target=enc(tgt), negatives=None if ns.k is None else enc(neg)), 0)
        score_cases.append(dict(mode=ns.mode.name, k=ns.k, user=reps[0], target=reps[1], negatives=reps[2], scores=scores, labels=labels,
                                loss=loss, temperature=0.05))

    # ---------------------------------------------------------------- configure_optimizers driven like Lightning drives it
    optim_cases = []
    for warmup, wd, clip, total, steps in ((3, 0.01, 1.0, 10, 8), (0.25, 0.0, None, 8, 6), (0, 0.1, 0.05, 5, 5)):
        m = make(NegativeSampleConfig(in_batch=True), temperature=0.05, learning_rate=1e-2, warmup_steps=warmup, weight_decay=wd)
        m.trainer = types.SimpleNamespace(estimated_stepping_batches=total)
        (opt,), (sch,) = m.configure_optimizers()
        sched = sch["scheduler"]
        groups = [[n for n, p in m.named_parameters() if any(p is q for q in grp["params"])] for grp in opt.param_groups]
        init = OrderedDict((n, p.detach().clone()) for n, p in m.named_parameters())
        gb = torch.Generator().manual_seed(77)
        rec = []
        for s in range(steps):
            b_seq, b_tgt = toy_batch(B, 9, vocab, gb), toy_batch(B, 5, vocab, gb)
            lr = [grp["lr"] for grp in opt.param_groups]
            loss = m.training_step(BatchSequenceWithNegative(sequence=enc(b_seq), target=enc(b_tgt), negatives=None), s)
            opt.zero_grad()
            loss.backward()
            grads = OrderedDict((n, p.grad.detach().clone()) for n, p in m.named_parameters())  # before clipping
            norm = None
            if clip is not None:
                norm = torch.nn.utils.clip_grad_norm_(m.parameters(), clip)
            opt.step()
            sched.step()
            rec.append(dict(lr=lr, loss=loss.detach(), grads=grads, grad_norm=norm, batch=(b_seq, b_tgt),
                            params=OrderedDict((n, p.detach().clone()) for n, p in m.named_parameters())))
        optim_cases.append(dict(warmup_steps=warmup, weight_decay=wd, gradient_clip_val=clip, estimated_stepping_batches=total, learning_rate=1e-2,
                                group_weight_decay=[grp["weight_decay"] for grp in opt.param_groups], group_names=groups, init=init, steps=rec,
                                betas=opt.defaults["betas"], eps=opt.defaults["eps"], temperature=0.05))
    # ---------------------------------------------------------------- the WHOLE training step of the reference on the real encoder family:
    # RecModule (reference) around transformers' RobertaModel (the library the reference delegates the encoder to; tiny config with the
    # true head size, seeded weights): in-batch negatives, cosine similarity, temperature 0.05 -> loss and d loss / d every parameter.
    from transformers import RobertaConfig, RobertaModel

    from oracle import ref_cpu as O

    cfg = O.EncoderConfig(hidden=128, heads=2, layers=2, intermediate=256, vocab=300, max_pos=130)
    hc = RobertaConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                       intermediate_size=cfg.intermediate, max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.token_type_size,
                       pad_token_id=cfg.pad_id, layer_norm_eps=cfg.ln_eps, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    sd = O.random_state_dict(O.roberta_param_shapes(cfg), seed=2025, std=0.05)

    class Wrapper(nn.Module):  # rec_retrieval/module/models/encoder/_base.py:32-49 with pooling_method="cls"
        def __init__(self):
            super().__init__()
            self.tokenizer = None
            self.model = RobertaModel(hc, add_pooling_layer=True)
            self.model.load_state_dict({k[len("model."):]: v for k, v in sd.items()}, strict=True)

        def forward(self, batch):
            return self.model(**batch).last_hidden_state[:, 0, :]

    def text_batch(B, L):
        lens = torch.randint(3, L + 1, (B,), generator=g)
        ids = torch.randint(3, cfg.vocab, (B, L), generator=g)
        ids[:, 0] = 0
        mask = (torch.arange(L).view(1, L) < lens.view(B, 1)).long()
        return {"input_ids": ids * mask + cfg.pad_id * (1 - mask), "attention_mask": mask}

    wrapper = Wrapper().train()
    rm = mod.RecModule(model=wrapper, evaluator=Evaluator(metrics=["NDCG"], ks=[1]), negative_sample=NegativeSampleConfig(in_batch=True),
                       similarity="cosine", temperature=0.05)
    seq_b, tgt_b = text_batch(10, 40), text_batch(10, 12)
    loss = rm.training_step(BatchSequenceWithNegative(sequence=enc(seq_b), target=enc(tgt_b), negatives=None), 0)
    loss.backward()
    grads = OrderedDict(("model." + k, (p.grad.detach().clone() if p.grad is not None else None)) for k, p in wrapper.model.named_parameters())
    step = dict(cfg=cfg.__dict__, state_dict=OrderedDict(("model." + k, v.detach().clone()) for k, v in wrapper.model.state_dict().items()),
                sequence=seq_b, target=tgt_b, loss=loss.detach(), grads=grads, temperature=0.05)
    torch.save(dict(scores=score_cases, optim=optim_cases, toy=dict(vocab=vocab, d=d, seed=5), roberta_step=step), OUT / "g9_finetune.pt")
    print("roberta step loss", float(loss))
    print("wrote", OUT / "g9_finetune.pt", [c["mode"] for c in score_cases], [len(c["steps"]) for c in optim_cases])


if __name__ == "__main__":
    main()
