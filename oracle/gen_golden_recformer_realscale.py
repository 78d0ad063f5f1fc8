#!/usr/bin/env python3
"""Real-scale accuracy fixture for the second model family (BUILD CONTAINER ONLY): tests/golden/g14_realscale_recformer_base.pt.

TEST INFRASTRUCTURE.  BASELINE configs[2] -- a 3-domain task-vector merge of Recformer-base (Longformer-base geometry: 12 x 768, 12 heads,
one-sided window 32, 4,098 positions, 4 token types, 51 item positions; 148 M parameters) with per-domain alpha -- run through the REFERENCE on
the CPU in fp32 on a Pantry-sized synthetic domain (4,968 items, 1,024 users with sequences up to 1,024 tokens):

  * the merge: the reference's ``load_merging_module(TASK_VECTOR, TASK_WISE)`` + ``load_weights_from_dict`` + ``get_state_dict()``
    (merge_test.py:35-71; imported through the PEP-695 -> 3.10 loader of oracle/gen_golden.py),
  * the encoder: the reference's own ``RecformerModel`` (encoder/recformer/models.py, loaded by path) driving transformers'
    LongformerEncoder exactly as oracle/gen_golden.py's G4 does (models.py:273-361, mask per :326-330), CLS row, ``F.normalize``,
  * scoring / loss / ranking / metrics as in oracle/gen_golden_8domain.py.

Inputs are regenerated from seeds by the test; stored: the first 256 users' U, every 8th row of E, labels (reference rank log-uniform in [1, 200]), the reference's
top-52 (indices, scores), label ranks with the reference scores three ranks either side, metrics, loss.
"""
from __future__ import annotations

import sys
import time
from collections import OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

N_ITEMS, N_USERS, U_KEEP, E_STRIDE, SEED_DOMAIN = 4968, 1024, 256, 8, 41000
SEED_PRE, SEED_FT = 3000, (3001, 3002, 3003)
ALPHAS = (0.42, 0.31, 0.27)
MAX_SEQ_LEN = 1024
# `python oracle/gen_golden_recformer_realscale.py large`: Recformer-LARGE (24 x 1,024, 16 heads, 435 M parameters: BASELINE configs[4]'s
# model) on a smaller sample (2,048 items, 512 users) -> tests/golden/g15_realscale_recformer_large.pt
LARGE = len(sys.argv) > 1 and sys.argv[1] == "large"
if LARGE:
    N_ITEMS, N_USERS, SEED_DOMAIN = 2048, 512, 42000
    SEED_PRE, SEED_FT = 4000, (4001, 4002, 4003)


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
    from transformers import LongformerConfig

    rm = GG.load_by_path("_ref_recformer_models", GG.REF / "rec_retrieval/module/models/encoder/recformer/models.py")
    cfg = O.EncoderConfig(max_pos=4098, token_type_size=4, max_item_embeddings=51, one_sided_window=32,
                          **(dict(hidden=1024, heads=16, layers=24, intermediate=4096) if LARGE else {}))
    t0 = time.time()
    hc = LongformerConfig(attention_window=[2 * cfg.one_sided_window] * cfg.layers, vocab_size=cfg.vocab, hidden_size=cfg.hidden,
                          num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
                          max_position_embeddings=cfg.max_pos, type_vocab_size=1, pad_token_id=cfg.pad_id, layer_norm_eps=cfg.ln_eps,
                          hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hc.token_type_size = cfg.token_type_size
    hc.max_item_embeddings = cfg.max_item_embeddings
    hc.pooler_type = "cls"

    class Wrapper(torch.nn.Module):  # models/_base.py BaseModel: state_dict keys 'model.<key>'
        def __init__(self):
            super().__init__()
            self.model = rm.RecformerModel(hc).eval()

        def forward(self, batch):  # recformer/models.py:273-361 with the mask built per :326-330 semantics (as G4)
            m = self.model
            am = m._merge_to_attention_mask(batch["attention_mask"], batch["global_attention_mask"])
            padding_len, input_ids, am, tt, pos, ip, _ = m._pad_to_window_size(
                input_ids=batch["input_ids"], attention_mask=am, token_type_ids=batch["token_type_ids"], position_ids=None,
                item_position_ids=batch["item_position_ids"], inputs_embeds=None, pad_token_id=m.config.pad_token_id)
            ext = (1.0 - am.to(torch.float32)) * torch.finfo(torch.float32).min
            emb = m.embeddings(input_ids=input_ids, position_ids=pos, item_position_ids=ip, token_type_ids=tt)
            enc = m.encoder(emb, attention_mask=ext, padding_len=padding_len, return_dict=True)
            return enc.last_hidden_state[:, 0]

    pre0 = O.random_state_dict(O.recformer_param_shapes(cfg), seed=SEED_PRE, std=0.02)
    w = Wrapper()
    assert set(w.state_dict().keys()) == set(pre0.keys()), "recformer key set mismatch"
    w.model.load_state_dict({k[len("model."):]: v for k, v in pre0.items()}, strict=True)
    pre = OrderedDict((k, v.detach().clone()) for k, v in w.state_dict().items())  # the reference model's key order
    fts = [O.perturbed_state_dict(pre, seed=s, std=1e-3) for s in SEED_FT]
    mm = load_merging_module(merge_type=MergeType.TASK_VECTOR, learn_type=LearnType.TASK_WISE, model=w, pretrain_state_dict=pre,
                             finetune_state_dicts=[dict(ft) for ft in fts], ignore_keys=set(), disable_softmax=True)
    mm.load_weights_from_dict({"global_weights": {"all": [1.0]}, "global_biases": {"all": [0.0]}, "per_weights": {"all": list(ALPHAS)}})
    merged = OrderedDict((k, v.detach().clone()) for k, v in mm.get_state_dict().items())
    del mm, w, fts
    model = Wrapper()
    model.model.load_state_dict({k[len("model."):]: v for k, v in merged.items()}, strict=True)
    print(f"3-way merge done in {time.time() - t0:.1f}s", flush=True)

    dom = make_domain("Pantry", N_ITEMS, N_USERS, 32, cfg.vocab, SEED_DOMAIN, kind="recformer", max_seq_len=MAX_SEQ_LEN)

    def encode(batches, key):
        outs = []
        with torch.no_grad():
            for i, b in enumerate(batches):
                outs.append(F.normalize(model(dict(getattr(b, key))), p=2, dim=-1))
                if i % 20 == 0:
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
    seq_lens = torch.cat([b.sequence["attention_mask"].sum(1) for b in dom.sequence_batches])
    out = dict(n_items=M, n_users=N_USERS, u_keep=U_KEEP, e_stride=E_STRIDE, seed_domain=SEED_DOMAIN, max_seq_len=MAX_SEQ_LEN, seed_pre=SEED_PRE, seed_ft=list(SEED_FT),
               alphas=list(ALPHAS), ft_std=1e-3, ks=ks, key_order=list(pre.keys()),
               pre_checksum=float(sum(v.double().sum() for v in pre.values() if v.is_floating_point())),
               merged_checksum=float(sum(v.double().sum() for v in merged.values() if v.is_floating_point())),
               U=U[:U_KEEP].clone(), E_rows=rows.to(torch.int32), E_sample=E[rows].clone(), E_checksum=float(E.double().sum()), labels=labels,
               ref_top52_idx=top.indices[:, :52].to(torch.int32).clone(), ref_top52_val=top.values[:, :52].clone(), label_rank=label_rank,
               label_score=lab_score.clone(), label_window=label_window, metrics={k: float(v) for k, v in metrics.items()}, loss=loss,
               longest_sequence=int(seq_lens.max()),
               versions=dict(torch=str(torch.__version__), transformers=str(__import__("transformers").__version__)))
    out["encoder"] = dict(hidden=cfg.hidden, heads=cfg.heads, layers=cfg.layers, intermediate=cfg.intermediate)
    path = ROOT / "tests" / "golden" / ("g15_realscale_recformer_large.pt" if LARGE else "g14_realscale_recformer_base.pt")
    torch.save(out, path)
    print("saved", path, path.stat().st_size, metrics, "loss", loss, "longest sequence", int(seq_lens.max()), f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
