#!/usr/bin/env python3
"""BLaIR-LARGE at true dimensions (BUILD CONTAINER ONLY): tests/golden/g16_realscale_blair_large.pt.

TEST INFRASTRUCTURE.  The reference's scripts offer ``--model_type blair_large`` (scripts/baselines/blair_base_*.sh: "You can change model_type to
blair_large"): RoBERTa-large geometry (24 x 1,024, 16 heads, 355 M parameters).  A 2-domain task-vector merge (per-domain alpha) through the
reference's ``load_merging_module`` / ``get_state_dict`` (merge_test.py:35-71), transformers' ``RobertaModel``, CLS pooled, ``F.normalize``,
``user @ item.T``, ``cross_entropy(scores / 0.05)`` and the reference's ``Evaluator`` on the CPU in fp32, on 2,048 items and 512 users.
Stored (as in oracle/gen_golden_recformer_realscale.py): the first 256 users' U, every 8th row of E, labels (reference rank log-uniform in
[1, 200]), the reference's top-52, label ranks with the reference scores three ranks either side, metrics, loss.
"""
from __future__ import annotations

import sys
import time
from collections import OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

N_ITEMS, N_USERS, U_KEEP, E_STRIDE, SEED_DOMAIN = 2048, 512, 256, 8, 43000
SEED_PRE, SEED_FT = 5000, (5001, 5002)
ALPHAS = (0.6, 0.4)
ENC = dict(hidden=1024, heads=16, layers=24, intermediate=4096)


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

    cfg = O.EncoderConfig(**ENC)
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
    pre = OrderedDict((k, v.detach().clone()) for k, v in w.state_dict().items())
    fts = [O.perturbed_state_dict(pre, seed=s, std=1e-3) for s in SEED_FT]
    mm = load_merging_module(merge_type=MergeType.TASK_VECTOR, learn_type=LearnType.TASK_WISE, model=w, pretrain_state_dict=pre,
                             finetune_state_dicts=[dict(ft) for ft in fts], ignore_keys=set(), disable_softmax=True)
    mm.load_weights_from_dict({"global_weights": {"all": [1.0]}, "global_biases": {"all": [0.0]}, "per_weights": {"all": list(ALPHAS)}})
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
                if i % 10 == 0:
                    print(f"  {key} batch {i}/{len(batches)}  {time.time() - t0:.0f}s", flush=True)
        return torch.cat(outs)

    E = encode(dom.item_batches, "items")
    U = encode(dom.sequence_batches, "sequence")
    scores = U @ E.T
    M = N_ITEMS
    top = torch.topk(scores, 200, dim=1)
    g = torch.Generator().manual_seed(SEED_DOMAIN + 7)
    pos = (torch.exp(torch.rand(N_USERS, generator=g) * torch.log(torch.tensor(200.0))).floor().long() - 1).clamp(0, 199)
    labels = top.indices[torch.arange(N_USERS), pos].clone()
    ks = [1, 5, 10, 50]
    metrics = dict(Evaluator(metrics=["NDCG", "RECALL"], ks=ks)(scores, labels, "test/"))
    loss = float(F.cross_entropy(scores / 0.05, labels))
    lab_score = scores[torch.arange(N_USERS), labels]
    label_rank = (scores > lab_score[:, None]).sum(1).to(torch.int32)
    srt = torch.sort(scores, dim=1, descending=True).values
    win = label_rank.long()[:, None] + torch.arange(-3, 4)[None, :]
    label_window = torch.where((win >= 0) & (win < M), srt.gather(1, win.clamp(0, M - 1)), torch.full(win.shape, float("nan")))
    rows = torch.arange(0, M, E_STRIDE)
    out = dict(n_items=M, n_users=N_USERS, u_keep=U_KEEP, e_stride=E_STRIDE, seed_domain=SEED_DOMAIN, seed_pre=SEED_PRE, seed_ft=list(SEED_FT),
               alphas=list(ALPHAS), ft_std=1e-3, ks=ks, key_order=list(pre.keys()), encoder=dict(ENC),
               pre_checksum=float(sum(v.double().sum() for v in pre.values())), merged_checksum=float(sum(v.double().sum() for v in merged.values())),
               U=U[:U_KEEP].clone(), E_rows=rows.to(torch.int32), E_sample=E[rows].clone(), E_checksum=float(E.double().sum()), labels=labels,
               ref_top52_idx=top.indices[:, :52].to(torch.int32).clone(), ref_top52_val=top.values[:, :52].clone(), label_rank=label_rank,
               label_score=lab_score.clone(), label_window=label_window, metrics={k: float(v) for k, v in metrics.items()}, loss=loss,
               versions=dict(torch=str(torch.__version__), transformers=str(__import__("transformers").__version__)))
    path = ROOT / "tests" / "golden" / "g16_realscale_blair_large.pt"
    torch.save(out, path)
    print("saved", path, path.stat().st_size, metrics, "loss", loss, f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
