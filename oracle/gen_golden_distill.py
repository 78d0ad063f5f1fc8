#!/usr/bin/env python3
"""Generate tests/golden/g7_distill_losses.pt from the reference's loss classes (BUILD CONTAINER ONLY).

TEST INFRASTRUCTURE, same rules as gen_golden.py: reads /root/reference at generation time, commits tensors only.
``rec_retrieval/module/recommender/loss_fn.py`` is loaded standalone by path (its package __init__ needs lightning, which
is absent); its only relative import, ``...merger.enums``, resolves through the reference importer of gen_golden.py.
Each case stores the inputs, the loss value and the reference autograd gradient d loss / d merged_logits.
"""
from __future__ import annotations

import sys
import types
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.gen_golden import OUT, REF, install_reference_importer, load_by_path  # noqa: E402


def main():
    torch.set_num_threads(4)
    install_reference_importer()
    # empty parent packages so the standalone module's relative import (three dots) has an anchor
    for name in ("rec_retrieval.module", "rec_retrieval.module.recommender"):
        pkg = types.ModuleType(name)
        pkg.__path__ = []  # namespace stub: nothing is executed from the reference's module/__init__.py
        sys.modules[name] = pkg
    import rec_retrieval.merger.enums as enums

    lf = load_by_path("rec_retrieval.module.recommender.loss_fn", REF / "rec_retrieval/module/recommender/loss_fn.py")
    LossType = enums.LossType
    T, COEF, MARGIN = 0.05, 1000.0, 0.1  # scripts/3_mergerec/*.sh: --temperature 0.05 --coefficient 1000
    makers = {
        "CE": lambda: lf.distill_loss_factory(LossType.CE),
        "KD": lambda: lf.distill_loss_factory(LossType.KD, temperature=T),
        "MSE": lambda: lf.distill_loss_factory(LossType.MSE),
        "ADAMERGING": lambda: lf.distill_loss_factory(LossType.ADAMERGING),
        "ADAMERGING_KD": lambda: lf.distill_loss_factory(LossType.ADAMERGING_KD, temperature=T, coefficient=COEF),
        "MERGED_PSEUDO_LABEL": lambda: lf.distill_loss_factory(LossType.MERGED_PSEUDO_LABEL),
        "SINGLE_PSEUDO_LABEL": lambda: lf.distill_loss_factory(LossType.SINGLE_PSEUDO_LABEL),
        "MERGED_PSEUDO_LABEL_KD": lambda: lf.distill_loss_factory(LossType.MERGED_PSEUDO_LABEL_KD, temperature=T, coefficient=COEF),
        "SINGLE_PSEUDO_LABEL_KD": lambda: lf.distill_loss_factory(LossType.SINGLE_PSEUDO_LABEL_KD, temperature=T, coefficient=COEF),
        "PAIRWISE": lambda: lf.DistillPairwiseLoss(MARGIN),
        "LISTNET": lambda: lf.DistillListNetLoss(T),
    }
    g = torch.Generator().manual_seed(777)
    inputs = []
    # cosine-similarity-like logits (|x| <= 1, the path's scores) and a wider-range case; ragged sizes; one Amazon-sized row
    for (n, m, scale) in [(1, 50, 0.3), (4, 333, 0.3), (16, 1000, 1.0), (3, 4097, 0.2), (2, 22855, 0.25)]:
        z = (torch.randn(n, m, generator=g) * scale).clamp(-1, 1)
        t = (z * 0.7 + torch.randn(n, m, generator=g) * scale * 0.5).clamp(-1, 1)
        inputs.append((z, t))
    cases = []
    for name, mk in makers.items():
        for z, t in inputs:
            zz = z.clone().requires_grad_(True)
            loss = mk()(zz, t)
            (grad,) = torch.autograd.grad(loss, zz)
            cases.append(dict(loss=name, z=z, t=t, value=loss.detach(), grad=grad, temperature=T, coefficient=COEF, margin=MARGIN))
    # the per-sample loop of DistillSequenceModule._forward_distill, restated call for call on the reference loss object
    d = 64
    item_embeddings = [torch.nn.functional.normalize(torch.randn(m, d, generator=g), dim=-1) for m in (120, 333)]
    seq_emb = [torch.nn.functional.normalize(torch.randn(40, d, generator=g), dim=-1) for _ in range(2)]
    score_embeddings = [s @ e.T for s, e in zip(seq_emb, item_embeddings)]
    reps = torch.nn.functional.normalize(torch.randn(16, d, generator=g), dim=-1)
    ds_idx = torch.randint(0, 2, (16,), generator=g).tolist()
    seq_ids = torch.randint(0, 40, (16,), generator=g).tolist()
    loss_obj = makers["SINGLE_PSEUDO_LABEL_KD"]()
    rr = reps.clone().requires_grad_(True)
    losses = []
    for i, (di, sid) in enumerate(zip(ds_idx, seq_ids)):
        logit = rr[i] @ item_embeddings[di].T
        losses.append(loss_obj(logit.unsqueeze(0), score_embeddings[di][sid].unsqueeze(0)))
    total = torch.stack(losses).mean()
    (rgrad,) = torch.autograd.grad(total, rr)
    fwd = dict(reps=reps, item_embeddings=item_embeddings, score_embeddings=score_embeddings, dataset_indexes=ds_idx, sequence_ids=seq_ids,
               value=total.detach(), rep_grad=rgrad, loss="SINGLE_PSEUDO_LABEL_KD", temperature=T, coefficient=COEF)
    torch.save(dict(cases=cases, forward_distill=fwd), OUT / "g7_distill_losses.pt")
    print("wrote", OUT / "g7_distill_losses.pt", len(cases), "loss cases")


if __name__ == "__main__":
    main()
