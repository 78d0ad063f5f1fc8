#!/usr/bin/env python3
"""Generate tests/golden/g20_merge_train_step_recformer_large.pt: ONE collaborative-merging optimisation step of the reference itself for BASELINE
configs[4]'s model -- Recformer-LARGE (24 x 1,024, 435 M parameters) with 4 domains (BUILD CONTAINER ONLY; 8 do not fit its memory); Recformer counterpart of
gen_golden_merge_train_realscale.py (g19): the encoder is the reference's own RecformerModel driven as in oracle/gen_golden.py's G4.

TEST INFRASTRUCTURE, same rules as gen_golden.py: reads /root/reference at generation time, commits tensors only.  The chain is the
reference's own: ``load_merging_module`` (TaskVectorMergingModule{TaskWise,LayerWise} via the PEP-695 importer of gen_golden.py) around a
wrapper of transformers' RobertaModel (12 x 768, 124.6 M parameters; 8 fine-tuned checkpoints = a 4 GB task-vector matrix under autograd) -> ``DistillSequenceModule.training_step`` (loaded standalone by path;
``lightning`` is absent, its ``LightningModule`` is replaced by a bare nn.Module with a no-op ``log``) with the reference's
``SinglePseudoLabelKDLoss`` -> ``loss.backward()`` through ``make_functional`` + torch autograd.  Recorded: inputs, loss and the gradients
of per_weights / global_weights / global_biases for both learn types.
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


def main():
    torch.set_num_threads(8)
    install_reference_importer()
    lightning = types.ModuleType("lightning")

    class LightningModule(nn.Module):
        def log(self, *a, **k):
            pass

    lightning.LightningModule = LightningModule
    sys.modules["lightning"] = lightning
    for name in ("rec_retrieval.module", "rec_retrieval.module.recommender", "rec_retrieval.module.distiller", "rec_retrieval.module.distiller.sequence"):
        pkg = types.ModuleType(name)
        pkg.__path__ = []
        sys.modules[name] = pkg
    import rec_retrieval.merger.enums as enums  # noqa: F401  (loss_fn's relative import)
    from transformers import BatchEncoding, LongformerConfig

    from rec_retrieval.merger.enums import LearnType, LossType, MergeType
    from rec_retrieval.merger.weight_learning import load_merging_module
    from rec_retrieval.types.model_batch import BatchDistillationSequence

    from oracle import ref_cpu as O

    lf = load_by_path("rec_retrieval.module.recommender.loss_fn", REF / "rec_retrieval/module/recommender/loss_fn.py")
    dm = load_by_path("rec_retrieval.module.distiller.sequence.module", REF / "rec_retrieval/module/distiller/sequence/module.py")

    rm = load_by_path("_ref_recformer_models", REF / "rec_retrieval/module/models/encoder/recformer/models.py")
    cfg = O.EncoderConfig(hidden=1024, heads=16, layers=24, intermediate=4096, max_pos=4098, token_type_size=4, max_item_embeddings=51, one_sided_window=32)
    hc = LongformerConfig(attention_window=[2 * cfg.one_sided_window] * cfg.layers, vocab_size=cfg.vocab, hidden_size=cfg.hidden,
                          num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
                          max_position_embeddings=cfg.max_pos, type_vocab_size=1, pad_token_id=cfg.pad_id, layer_norm_eps=cfg.ln_eps,
                          hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hc.token_type_size = cfg.token_type_size
    hc.max_item_embeddings = cfg.max_item_embeddings
    hc.pooler_type = "cls"
    base_sd = O.random_state_dict(O.recformer_param_shapes(cfg), seed=7001, std=0.02)

    class Wrapper(nn.Module):  # models/_base.py: state_dict keys 'model.<key>'; recformer/models.py:273-361 with the mask built per :326-330 (as G4)
        def __init__(self):
            super().__init__()
            self.model = rm.RecformerModel(hc)
            self.model.load_state_dict({k[len("model."):]: v for k, v in base_sd.items()}, strict=True)

        def forward(self, batch):
            m = self.model
            am = m._merge_to_attention_mask(batch["attention_mask"], batch["global_attention_mask"])
            padding_len, input_ids, am, tt, pos, ip, _ = m._pad_to_window_size(
                input_ids=batch["input_ids"], attention_mask=am, token_type_ids=batch["token_type_ids"], position_ids=None,
                item_position_ids=batch["item_position_ids"], inputs_embeds=None, pad_token_id=m.config.pad_token_id)
            ext = (1.0 - am.to(torch.float32)) * torch.finfo(torch.float32).min
            emb = m.embeddings(input_ids=input_ids, position_ids=pos, item_position_ids=ip, token_type_ids=tt)
            enc = m.encoder(emb, attention_mask=ext, padding_len=padding_len, return_dict=True)
            return enc.last_hidden_state[:, 0]

    probe = Wrapper()
    pre = OrderedDict(("model." + k, v.detach().clone()) for k, v in probe.model.state_dict().items())
    N = 4   # (eight 1.7 GB checkpoints + their task-vector matrix + autograd exceed this container's 62 GB: four domains)
    fts = [O.perturbed_state_dict(pre, seed=7100 + i, std=1e-3) for i in range(N)]
    g = torch.Generator().manual_seed(299)
    B, L, d = 16, 48, cfg.hidden   # merge_train's batch: 16 pseudo users = item texts of ~40 tokens
    lens = torch.randint(3, L + 1, (B,), generator=g)
    ids = torch.randint(3, cfg.vocab, (B, L), generator=g)
    ids[:, 0] = 0
    mask = (torch.arange(L).view(1, L) < lens.view(B, 1)).long()
    ids = ids * mask + cfg.pad_id * (1 - mask)
    pos_ = torch.arange(L).view(1, L).expand(B, L)
    tt = torch.where(((pos_ - 1) % 38) < 3, torch.ones_like(ids), torch.full_like(ids, 2))
    tt[:, 0] = 0
    tt = torch.where(mask.bool(), tt, torch.full_like(tt, 3))          # collate pads (recformer_utils.py:97,99)
    ip = torch.clamp(1 + (pos_ - 1) // 38, max=50)
    ip[:, 0] = 0
    ip = torch.where(mask.bool(), ip, torch.zeros_like(ip))
    gm = torch.zeros_like(ids)
    gm[:, 0] = 1
    sizes = (4968, 12101, 18357, 11924)   # four of the eight catalogs
    items = [torch.nn.functional.normalize(torch.randn(m, d, generator=g), dim=-1) for m in sizes]
    teachers = [torch.randn(B, m, generator=g).clamp(-1, 1) for m in sizes]
    ds_idx = [i % N for i in range(B)]
    seq_ids = list(range(B))
    T, COEF = 0.05, 1000.0
    cases = []
    # one learn type per process (argv[1]): two merging modules of this size do not fit the container's memory together; the second run
    # appends its case to the file the first wrote
    for learn in (sys.argv[1],):
        w = Wrapper()
        mm = load_merging_module(merge_type=MergeType.TASK_VECTOR, learn_type=LearnType[learn], model=w, pretrain_state_dict=w.state_dict(),
                                 finetune_state_dicts=[dict(ft) for ft in fts], ignore_keys=set(), disable_softmax=True, initial_per_weight=0.3)
        module = dm.DistillSequenceModule(merged_model=mm, score_embeddings=teachers,
                                          loss_fn=lf.distill_loss_factory(LossType.SINGLE_PSEUDO_LABEL_KD, temperature=T, coefficient=COEF),
                                          similarity="cosine")
        module.item_embeddings = items
        batch = BatchDistillationSequence(dataset_indexes=ds_idx, sequence_ids=seq_ids, sequence=BatchEncoding({"input_ids": ids, "attention_mask": mask, "token_type_ids": tt, "item_position_ids": ip, "global_attention_mask": gm}))
        loss = module.training_step(batch, 0)
        loss.backward()
        grads = {name: OrderedDict((k, p.grad.detach().clone() if p.grad is not None else None) for k, p in getattr(mm, name).items())
                 for name in ("per_weights", "global_weights", "global_biases")}
        cases.append(dict(learn_type=learn, loss=loss.detach(), grads=grads, groups=list(mm.per_weights.keys())))
        print(learn, float(loss), {k: v.tolist() for k, v in list(grads["per_weights"].items())[:2]}, flush=True)
        del mm, module, w
    # the fine-tuned checkpoints are regenerated by the tests from the oracle's seeded perturbation (checksum guards drift)
    # inputs that are cheap to regenerate stay out of the fixture: the pretrained weights (seed + key order), the item matrices and the
    # teacher scores (one generator, in the order above)
    torch.save(dict(cfg=cfg.__dict__, pretrain_seed=7001, pretrain_std=0.02, key_order=list(pre.keys()), pretrain_checksum=float(sum(v.double().sum() for v in pre.values() if v.is_floating_point())),
                    catalog_sizes=list(sizes), data_seed=299, batch=(B, L), finetune_seeds=[7100 + i for i in range(N)], finetune_std=1e-3,
                    finetune_checksum=float(sum(v.double().sum() for ft in fts for v in ft.values() if v.is_floating_point())), input_ids=ids, attention_mask=mask, token_type_ids=tt, item_position_ids=ip, global_attention_mask=gm,
                    item_checksum=float(sum(x.double().sum() for x in items)), teacher_checksum=float(sum(x.double().sum() for x in teachers)),
                    dataset_indexes=ds_idx, sequence_ids=seq_ids, temperature=T, coefficient=COEF, initial_per_weight=0.3, cases=cases),
               OUT / f"g20_part_{sys.argv[1]}.pt")
    parts = [OUT / f"g20_part_{lt}.pt" for lt in ("TASK_WISE", "LAYER_WISE")]
    if all(p.exists() for p in parts):
        a, b = (torch.load(p, weights_only=False) for p in parts)
        a["cases"] = a["cases"] + b["cases"]
        torch.save(a, OUT / "g20_merge_train_step_recformer_large.pt")
        for p in parts:
            p.unlink()
    print("wrote", OUT / "g20_merge_train_step_recformer_large.pt")


if __name__ == "__main__":
    main()
