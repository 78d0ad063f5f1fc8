#!/usr/bin/env python3
"""Eight-domain accuracy fixture (BUILD CONTAINER ONLY): tests/golden/g13_8domain_blair_base.pt.

TEST INFRASTRUCTURE.  north_star's clause "NDCG@10 within 1e-3 of reference across all 8 Amazon domains", for BASELINE configs[3]'s model:
ONE 8-domain task-vector merge of BLaIR-base at true dimensions (the reference's ``load_merging_module(TASK_VECTOR, TASK_WISE)`` +
``load_weights_from_dict`` + ``get_state_dict()``, merge_test.py:35-71, fixed per-domain alpha), evaluated the way merge_test.py evaluates
it -- on EVERY domain's full catalog (mergerec_amd.synthetic.CATALOG_SIZES: 4,968 ... 27,932 items, 114,075 in all) with 4,096 test users
per domain: transformers' RobertaModel (the arithmetic the reference delegates to, models/_base.py:56, encoder/_base.py:37-45), CLS pooled,
``F.normalize`` (module.py:74-77), ``user @ item.T`` (module.py:137), ``cross_entropy(scores / 0.05)`` (module.py:356), the reference's
``Evaluator`` (evaluator.py:31-49, metrics.py:38-88) as imported -- all on the CPU in fp32.

Inputs are regenerated from seeds by the test; per domain the fixture stores the first 256 users' embeddings U (256, 768), every 64th row of the item
matrix E (for the logit check), the labels (the item the REFERENCE ranks at a log-uniform position in [1, 200]: NDCG@10 near 0.3 and
sensitive to rank changes around every cutoff), the reference's top-50 (indices and scores), the scores of ranks 51 and 52 (near-tie
bookkeeping at the cutoff), per-user label ranks with the reference scores three ranks either side of the label, the metric dict and the loss.
"""
from __future__ import annotations

import sys
import time
from collections import OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

N_USERS, U_KEEP, E_STRIDE = 4096, 256, 64
# With 1,024 users ONE label crossing a Recall cut-off through a verified near-tie moves that metric by 9.8e-4 -- the whole 1e-3 bound; with
# 4,096 users it is 2.4e-4.  r04: EVERY domain is evaluated on 4,096 users (r03 had widened only the five domains where the crossing had been
# observed -- a fixture tuned per domain) and every entry is written by THIS revision of the script (17-column label windows).
# `python oracle/gen_golden_8domain.py <names>` regenerates a subset into g13_partial.pt and `--merge` folds it into the fixture.
LABEL_WIN = 8  # reference scores kept either side of the label's rank
SEED_PRE, SEED_FT = 2000, tuple(range(2001, 2009))
ALPHAS = (0.30, 0.10, 0.20, 0.15, 0.05, 0.25, 0.10, 0.20)
SEED_DOMAIN0 = 31000


def merge_partial():
    import torch

    full_p, part_p = ROOT / "tests" / "golden" / "g13_8domain_blair_base.pt", ROOT / "tests" / "golden" / "g13_partial.pt"
    full, part = torch.load(full_p, weights_only=False), torch.load(part_p, weights_only=False)
    for key in ("seed_pre", "seed_ft", "alphas", "ft_std", "ks", "key_order", "pre_checksum", "merged_checksum"):
        assert full[key] == part[key], key
    for name, d in part["domains"].items():
        full["domains"][name] = d
    for name, d in full["domains"].items():
        d.setdefault("n_users", full["n_users"])
    full["n_users"] = part["n_users"]
    torch.save(full, full_p)
    part_p.unlink()
    print("merged", list(part["domains"]), "->", full_p, full_p.stat().st_size)


def main():
    import torch
    import torch.nn.functional as F

    import gen_golden as GG

    torch.set_num_threads(int(__import__("os").environ.get("GEN_THREADS", "8")))
    GG.install_reference_importer()
    from oracle import ref_cpu as O
    from mergerec_amd.synthetic import CATALOG_SIZES, make_domain

    from rec_retrieval.evaluator import Evaluator
    from rec_retrieval.merger.enums import LearnType, MergeType
    from rec_retrieval.merger.weight_learning import load_merging_module
    from transformers import RobertaConfig, RobertaModel

    only = sys.argv[1].split(",") if len(sys.argv) > 1 else None  # subset of domains (smoke runs)
    cfg = O.EncoderConfig()
    t0 = time.time()
    pre = O.random_state_dict(O.roberta_param_shapes(cfg), seed=SEED_PRE, std=0.02)
    hc = RobertaConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                       intermediate_size=cfg.intermediate, max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.token_type_size,
                       pad_token_id=cfg.pad_id, layer_norm_eps=cfg.ln_eps, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)

    class Wrapper(torch.nn.Module):  # models/_base.py BaseModel: state_dict keys 'model.<hf-key>'
        def __init__(self):
            super().__init__()
            self.model = RobertaModel(hc, add_pooling_layer=True).eval()

        def forward(self, batch):
            return self.model(**batch).last_hidden_state[:, 0, :]

    w = Wrapper()
    w.model.load_state_dict({k[len("model."):]: v for k, v in pre.items()}, strict=True)
    pre = OrderedDict((k, v.detach().clone()) for k, v in w.state_dict().items())  # the installed library's key order
    fts = [O.perturbed_state_dict(pre, seed=s, std=1e-3) for s in SEED_FT]
    mm = load_merging_module(merge_type=MergeType.TASK_VECTOR, learn_type=LearnType.TASK_WISE, model=w, pretrain_state_dict=pre,
                             finetune_state_dicts=[dict(ft) for ft in fts], ignore_keys=set(), disable_softmax=True)
    mm.load_weights_from_dict({"global_weights": {"all": [1.0]}, "global_biases": {"all": [0.0]}, "per_weights": {"all": list(ALPHAS)}})
    merged = OrderedDict((k, v.detach().clone()) for k, v in mm.get_state_dict().items())
    del mm, w, fts
    model = Wrapper()
    model.model.load_state_dict({k[len("model."):]: v for k, v in merged.items()}, strict=True)
    print(f"8-way merge done in {time.time() - t0:.1f}s", flush=True)

    def encode(batches, key, tag):
        outs = []
        with torch.no_grad():
            for i, b in enumerate(batches):
                enc = getattr(b, key)
                outs.append(F.normalize(model({"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"]}), p=2, dim=-1))
                if i % 50 == 0:
                    print(f"  {tag} {key} batch {i}/{len(batches)}  {time.time() - t0:.0f}s", flush=True)
        return torch.cat(outs)

    ks = [1, 5, 10, 50]
    domains = OrderedDict()
    for d, (name, M) in enumerate(CATALOG_SIZES.items()):
        if only and name not in only:
            continue
        seed = SEED_DOMAIN0 + d
        dom = make_domain(name, M, N_USERS, 32, cfg.vocab, seed)
        E = encode(dom.item_batches, "items", name)
        U = encode(dom.sequence_batches, "sequence", name)
        scores = U @ E.T
        top = torch.topk(scores, 200, dim=1)
        g = torch.Generator().manual_seed(seed + 7)
        pos = (torch.exp(torch.rand(N_USERS, generator=g) * torch.log(torch.tensor(200.0))).floor().long() - 1).clamp(0, 199)
        labels = top.indices[torch.arange(N_USERS), pos].clone()
        metrics = dict(Evaluator(metrics=["NDCG", "RECALL"], ks=ks)(scores, labels, "test/"))
        loss = float(F.cross_entropy(scores / 0.05, labels))
        lab_score = scores[torch.arange(N_USERS), labels]
        label_rank = (scores > lab_score[:, None]).sum(1).to(torch.int32)
        srt = torch.sort(scores, dim=1, descending=True).values
        win = label_rank.long()[:, None] + torch.arange(-LABEL_WIN, LABEL_WIN + 1)[None, :]  # sorted positions rank - w .. rank + w
        label_window = torch.where((win >= 0) & (win < M), srt.gather(1, win.clamp(0, M - 1)), torch.full(win.shape, float("nan")))
        rows = torch.arange(0, M, E_STRIDE)
        domains[name] = dict(n_items=M, n_users=N_USERS, seed=seed, U=U[:U_KEEP].clone(), E_rows=rows.to(torch.int32), E_sample=E[rows].clone(),
                             E_checksum=float(E.double().sum()), labels=labels, ref_top52_idx=top.indices[:, :52].to(torch.int32).clone(),
                             ref_top52_val=top.values[:, :52].clone(), label_rank=label_rank, label_score=lab_score.clone(), label_window=label_window,
                             metrics={k: float(v) for k, v in metrics.items()}, loss=loss)
        print(f"{name}: M={M} NDCG@10={metrics['test/NDCG@10']:.4f} loss={loss:.4f}  {time.time() - t0:.0f}s", flush=True)

    out = dict(n_users=N_USERS, u_keep=U_KEEP, e_stride=E_STRIDE, seed_pre=SEED_PRE, seed_ft=list(SEED_FT), alphas=list(ALPHAS), ft_std=1e-3, ks=ks,
               key_order=list(pre.keys()), pre_checksum=float(sum(v.double().sum() for v in pre.values())),
               merged_checksum=float(sum(v.double().sum() for v in merged.values())), domains=domains,
               versions=dict(torch=str(torch.__version__), transformers=str(__import__("transformers").__version__)))
    path = ROOT / "tests" / "golden" / ("g13_8domain_blair_base.pt" if not only else "g13_partial.pt")
    torch.save(out, path)
    print("saved", path, path.stat().st_size, f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    merge_partial() if "--merge" in sys.argv else main()
