#!/usr/bin/env python3
"""Real-scale accuracy fixture (BUILD CONTAINER ONLY): tests/golden/g12_realscale_blair_base.pt.

TEST INFRASTRUCTURE.  Runs the REFERENCE's own pipeline for BASELINE configs[1] (2-domain merge, fixed alpha = 0.5) at
BLaIR-base true dimensions on a Pantry-sized synthetic domain (4,968 items, 2,048 test users) on the CPU, in fp32:

  * the merge: the reference's ``load_merging_module(TASK_VECTOR, TASK_WISE)`` + ``load_weights_from_dict`` +
    ``get_state_dict()`` (merge_test.py:35-71; imported through the PEP-695 -> 3.10 loader of oracle/gen_golden.py),
  * the encoder: transformers' ``RobertaModel`` (the third-party arithmetic the reference delegates to,
    models/_base.py:56, encoder/_base.py:37-45), CLS pooled, ``F.normalize`` (module.py:74-77),
  * scoring: ``user @ item.T`` (module.py:137), ``cross_entropy(scores / 0.05)`` (module.py:356),
  * ranking / metrics: the reference's ``Evaluator`` (evaluator.py:31-49, metrics.py:38-88) as imported.

Inputs are regenerated from seeds by the test (weights: ``ref_cpu.random_state_dict`` / ``perturbed_state_dict``;
token ids: ``mergerec_amd.synthetic.make_domain``), so the fixture stores only OUTPUTS plus the labels: E (4968, 768),
U (2048, 768), the reference's top-50, per-user label ranks and the metric dict.

Labels: a random-weight encoder scores a random label at a random rank (NDCG@10 ~ 1e-3: any comparison would pass).
Each user's label is therefore the item the REFERENCE ranks at a log-uniform position in [1, 200], which puts
NDCG@10 near 0.3 and makes it sensitive to rank changes around every cutoff.

The oracle restatement (oracle/ref_cpu.py) is run beside the reference on the first 256 users / 512 items and must agree
to 2e-6 -- recorded in the fixture as ``oracle_vs_reference``.
"""
from __future__ import annotations

import sys
import time
from collections import OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

N_ITEMS, N_USERS, SEED_DOMAIN = 4968, 2048, 20260
SEED_PRE, SEED_FT = 1000, (1001, 1002)
ALPHA = 0.5


def main():
    import torch
    import torch.nn.functional as F

    import gen_golden as GG

    torch.set_num_threads(8)
    GG.install_reference_importer()
    from oracle import ref_cpu as O
    from mergerec_amd.synthetic import make_domain

    from rec_retrieval.evaluator import Evaluator
    from rec_retrieval.merger.enums import LearnType, MergeType
    from rec_retrieval.merger.weight_learning import load_merging_module
    from transformers import RobertaConfig, RobertaModel

    cfg = O.EncoderConfig()  # BLaIR-base: 12 x 768, 12 heads, vocab 50265, 514 positions
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
    mm.load_weights_from_dict({"global_weights": {"all": [1.0]}, "global_biases": {"all": [0.0]}, "per_weights": {"all": [ALPHA, ALPHA]}})
    merged = OrderedDict((k, v.detach().clone()) for k, v in mm.get_state_dict().items())
    del mm, w, fts
    model = Wrapper()
    model.model.load_state_dict({k[len("model."):]: v for k, v in merged.items()}, strict=True)
    print(f"merge done in {time.time() - t0:.1f}s", flush=True)

    dom = make_domain("Pantry", N_ITEMS, N_USERS, 32, cfg.vocab, SEED_DOMAIN)

    def encode(batches, key):
        outs = []
        with torch.no_grad():
            for i, b in enumerate(batches):
                enc = getattr(b, key)
                outs.append(F.normalize(model({"input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"]}), p=2, dim=-1))
                if i % 20 == 0:
                    print(f"  {key} batch {i}/{len(batches)}  {time.time() - t0:.0f}s", flush=True)
        return torch.cat(outs)

    E = encode(dom.item_batches, "items")
    U = encode(dom.sequence_batches, "sequence")
    scores = U @ E.T
    top = torch.topk(scores, 200, dim=1)
    g = torch.Generator().manual_seed(SEED_DOMAIN + 1)
    pos = (torch.exp(torch.rand(N_USERS, generator=g) * torch.log(torch.tensor(200.0))).floor().long() - 1).clamp(0, 199)
    labels = top.indices[torch.arange(N_USERS), pos].clone()
    ks = [1, 5, 10, 50]
    metrics = dict(Evaluator(metrics=["NDCG", "RECALL"], ks=ks)(scores, labels, "test/"))
    loss = float(F.cross_entropy(scores / 0.05, labels))
    lab_score = scores[torch.arange(N_USERS), labels]
    label_rank = (scores > lab_score[:, None]).sum(1).to(torch.int32)  # number of strictly greater scores
    # smallest gap between a label's score and any other item's score, and between consecutive top-51 scores: the near-tie budget
    top51 = torch.topk(scores, 51, dim=1).values
    min_gap_top = float((top51[:, :-1] - top51[:, 1:]).min())

    # the restatement beside the reference on a slice
    osd = merged
    nu, ni = 256, 512
    with torch.no_grad():
        cat = lambda bs, key, n: {k: torch.nn.utils.rnn.pad_sequence([r for b in bs for r in getattr(b, key)[k]][:n], batch_first=True,
                                                                       padding_value=(1 if k == "input_ids" else 0)) for k in ("input_ids", "attention_mask")}
        ub, ib = cat(dom.sequence_batches, "sequence", nu), cat(dom.item_batches, "items", ni)
        Uo = O.maybe_normalize(torch.cat([O.roberta_encode(osd, ub["input_ids"][s:s + 32], ub["attention_mask"][s:s + 32], cfg, "model.") for s in range(0, nu, 32)]))
        Eo = O.maybe_normalize(torch.cat([O.roberta_encode(osd, ib["input_ids"][s:s + 64], ib["attention_mask"][s:s + 64], cfg, "model.") for s in range(0, ni, 64)]))
    ovr = dict(users=nu, items=ni, user_max_abs_diff=float((Uo - U[:nu]).abs().max()), item_max_abs_diff=float((Eo - E[:ni]).abs().max()),
               logit_max_abs_diff=float((Uo @ Eo.T - scores[:nu, :ni]).abs().max()))
    print("oracle vs reference:", ovr, flush=True)
    assert ovr["logit_max_abs_diff"] < 2e-6, ovr

    out = dict(
        n_items=N_ITEMS, n_users=N_USERS, seed_domain=SEED_DOMAIN, seed_pre=SEED_PRE, seed_ft=list(SEED_FT), alpha=ALPHA, ft_std=1e-3,
        key_order=list(pre.keys()), pre_checksum=float(sum(v.double().sum() for v in pre.values())),
        merged_checksum=float(sum(v.double().sum() for v in merged.values())),
        E=E.clone(), U=U.clone(), labels=labels, ref_top50_idx=top.indices[:, :50].to(torch.int32).clone(), ref_top50_val=top.values[:, :50].clone(),
        label_rank=label_rank, label_pos_drawn=pos.to(torch.int32), metrics=metrics, loss=loss, ks=ks, min_gap_top51=min_gap_top,
        oracle_vs_reference=ovr,
        versions=dict(torch=str(torch.__version__), transformers=str(__import__("transformers").__version__)),
    )
    path = ROOT / "tests" / "golden" / "g12_realscale_blair_base.pt"
    torch.save(out, path)
    print("saved", path, path.stat().st_size, metrics, "loss", loss, "min top gap", min_gap_top, f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
