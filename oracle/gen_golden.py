#!/usr/bin/env python3
"""Generate tests/golden/*.pt from the runnable subset of the reference (BUILD CONTAINER ONLY).

TEST INFRASTRUCTURE.  Reads /root/reference at generation time and never copies its source: what
is committed are input/output tensors only.  /root/reference does not exist on the GPU box, so
nothing in tests/, bench.py or smoke() imports this file.

How the reference is run here (python 3.10.12, torch 2.10, transformers 5.15.0):
  * ``rec_retrieval.evaluator`` imports as-is.
  * ``rec_retrieval.merger`` (ModelMerger, algorithms, weight_learning incl. ``load_merging_module``
    and the TaskWise/LayerWise modules) uses PEP-695 syntax (``type X = ...``, ``class C[T](...)``)
    that python 3.10 cannot parse.  A meta-path loader restricted to /root/reference rewrites
    exactly those two constructs in memory (``type X = Y`` -> ``X = Y``; ``class C[T](B)`` ->
    ``class C(B)``) before compiling; no arithmetic is touched.
  * ``rec_retrieval.module.*`` needs lightning/peft/tyro (absent) and cannot be imported; the
    Recformer model file is loaded standalone by path (it imports only torch + transformers) and
    its RecformerEmbeddings / _merge_to_attention_mask / _pad_to_window_size are called directly,
    driving the library's LongformerEncoder with the ``(1 - mask) * finfo.min`` mask of
    recformer/models.py:326-330 (the library helper's signature changed in transformers 5.x).
  * The RoBERTa encoder arithmetic lives in the third-party ``transformers`` package
    (requirements.txt pins ~=4.51.3; 5.15.0 is what is installed): RobertaModel built from local
    configs with seeded random weights is the golden source for a8/a11.
"""
from __future__ import annotations

import importlib.abc
import importlib.machinery
import importlib.util
import os
import re
import sys
from collections import OrderedDict
from pathlib import Path

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parent.parent / "tests" / "golden"

# --------------------------------------------------------------------------------------------
# import hook: PEP-695 -> 3.10 syntax, for files under /root/reference only
# --------------------------------------------------------------------------------------------
_TYPE_ALIAS = re.compile(r"^type\s+(\w+)\s*=", re.M)
_GENERIC_CLASS = re.compile(r"^class\s+(\w+)\[\w+\]\(", re.M)


class _DowngradeLoader(importlib.machinery.SourceFileLoader):
    def source_to_code(self, data, path, *, _optimize=-1):
        text = data.decode("utf-8") if isinstance(data, (bytes, bytearray)) else data
        text = _TYPE_ALIAS.sub(r"\1 =", text)
        text = _GENERIC_CLASS.sub(r"class \1(", text)
        return compile(text, path, "exec", dont_inherit=True, optimize=_optimize)

    def get_code(self, fullname):  # never read/write .pyc next to the read-only reference
        path = self.get_filename(fullname)
        return self.source_to_code(self.get_data(path), path)


class _RefFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path, target=None):
        if not fullname.startswith("rec_retrieval"):
            return None
        parts = fullname.split(".")
        base = REF.joinpath(*parts)
        if base.is_dir() and (base / "__init__.py").exists():
            loc = base / "__init__.py"
            return importlib.util.spec_from_file_location(
                fullname, loc, loader=_DowngradeLoader(fullname, str(loc)), submodule_search_locations=[str(base)]
            )
        loc = base.with_suffix(".py")
        if loc.exists():
            return importlib.util.spec_from_file_location(fullname, loc, loader=_DowngradeLoader(fullname, str(loc)))
        return None


def install_reference_importer():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _RefFinder())


def load_by_path(name: str, path: Path):
    spec = importlib.util.spec_from_file_location(name, path, loader=_DowngradeLoader(name, str(path)))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


# --------------------------------------------------------------------------------------------
def main():
    import torch

    torch.set_num_threads(4)
    install_reference_importer()
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from oracle import ref_cpu as O  # only for deterministic synthetic weights / shapes

    OUT.mkdir(parents=True, exist_ok=True)

    # ------------------------------------------------------------------ G1 evaluator (a18, a19)
    from rec_retrieval.evaluator import Evaluator

    g = torch.Generator().manual_seed(1234)
    cases = []
    for ci, (U, M, ties) in enumerate([(64, 300, True), (37, 120, False), (5, 50, False)]):
        scores = torch.randn(U, M, generator=g)
        labels = torch.randint(0, M, (U,), generator=g)
        if ties:
            # exact ties among non-label items (so the metrics do not depend on torch.topk's
            # unspecified tie order), some inside the top-50, plus duplicated rows of maxima
            for u in range(0, U, 3):
                cols = [c for c in torch.randperm(M, generator=g)[:6].tolist() if c != labels[u].item()]
                scores[u, cols] = float(scores[u, cols[0]])
            for u in range(1, U, 5):
                cols = [c for c in torch.randperm(M, generator=g)[:4].tolist() if c != labels[u].item()]
                scores[u, cols] = float(scores[u].max()) + 1.0
            # make some labels rank high so NDCG is non-trivial
            for u in range(0, U, 2):
                scores[u, labels[u]] = float(scores[u].max()) + 0.5 + 0.01 * u
        ks = [1, 5, 10, 50]
        ev = Evaluator(metrics=["NDCG", "RECALL"], ks=ks)
        metrics = ev(scores, labels, "test/")
        topk = torch.topk(scores, min(50, M), dim=1)
        cases.append(dict(scores=scores, labels=labels, ks=ks, metrics=dict(metrics), ref_topk_idx=topk.indices,
                          ref_topk_val=topk.values, metric_key_order=list(metrics.keys())))
    torch.save(cases, OUT / "g1_evaluator.pt")
    print("G1 ok", cases[0]["metrics"])

    # ------------------------------------------------------------------ HF tiny RoBERTa (G3) and wrapper for G2/G5
    from transformers import RobertaConfig, RobertaModel, BatchEncoding

    def hf_roberta(cfg: "O.EncoderConfig", sd):
        hc = RobertaConfig(
            vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
            intermediate_size=cfg.intermediate, max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.token_type_size,
            pad_token_id=cfg.pad_id, layer_norm_eps=cfg.ln_eps, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
        )
        m = RobertaModel(hc, add_pooling_layer=True).eval()
        missing = m.load_state_dict({k[len("model."):]: v for k, v in sd.items()}, strict=True)
        return m

    tiny = O.EncoderConfig(hidden=128, heads=2, layers=2, intermediate=256, vocab=200, max_pos=66)  # head_dim 64 like the real models
    tiny_sd = O.random_state_dict(O.roberta_param_shapes(tiny), seed=1000, std=0.2)
    m = hf_roberta(tiny, tiny_sd)
    # transformers 5.x orders the embedding tensors (word, token_type, LayerNorm, position); 4.51.3 (the
    # reference's pin) orders them (word, position, token_type, LayerNorm).  The flat-vector order is
    # whatever the pretrained wrapper's state_dict() yields (_factory.py:55-66), so the fixture keeps
    # the installed library's order and the build must be order-agnostic.
    tiny_sd = OrderedDict(("model." + k, tiny_sd["model." + k]) for k in m.state_dict().keys())
    B, L = 5, 23
    ids = torch.randint(3, tiny.vocab, (B, L), generator=g)
    lens = [23, 1, 7, 16, 2]
    mask = torch.zeros(B, L, dtype=torch.int64)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
        ids[b, 0] = 0
        if n > 1:
            ids[b, n - 1] = 2
        ids[b, n:] = tiny.pad_id
    with torch.no_grad():
        out = m(input_ids=ids, attention_mask=mask, output_hidden_states=True)
    g3 = dict(
        cfg=tiny.__dict__, state_dict=tiny_sd, input_ids=ids, attention_mask=mask,
        hidden_states=[h.clone() for h in out.hidden_states], cls=out.last_hidden_state[:, 0].clone(),
    )
    # true-dim single layer (weights regenerated from the seed by the test; checksum guards drift)
    big = O.EncoderConfig(hidden=768, heads=12, layers=1, intermediate=3072, vocab=1000, max_pos=514)
    big_sd = O.random_state_dict(O.roberta_param_shapes(big), seed=1001, std=0.02)
    mb = hf_roberta(big, big_sd)
    Bb, Lb = 3, 40
    idb = torch.randint(3, big.vocab, (Bb, Lb), generator=g)
    mkb = torch.ones(Bb, Lb, dtype=torch.int64)
    idb[:, 0] = 0
    mkb[1, 29:] = 0
    idb[1, 29:] = 1
    mkb[2, 5:] = 0
    idb[2, 5:] = 1
    with torch.no_grad():
        ob = mb(input_ids=idb, attention_mask=mkb, output_hidden_states=True)
    g3["big"] = dict(
        cfg=big.__dict__, seed=1001, std=0.02, checksum=float(sum(v.double().sum() for v in big_sd.values())),
        input_ids=idb, attention_mask=mkb, emb=ob.hidden_states[0].clone(), last=ob.last_hidden_state.clone(),
    )
    torch.save(g3, OUT / "g3_roberta.pt")
    print("G3 ok", g3["cls"][0, :4])

    # ------------------------------------------------------------------ G2/G5 merger via the reference's own classes
    from rec_retrieval.merger import ModelMerger
    from rec_retrieval.merger.enums import MergeType, LearnType
    from rec_retrieval.merger.weight_learning import load_merging_module
    from rec_retrieval.merger.algorithms.task_vector import get_task_vectors

    class Wrapper(torch.nn.Module):  # stands for models/_base.py BaseModel: state_dict keys 'model.<hf-key>'
        def __init__(self, hf):
            super().__init__()
            self.model = hf

        def forward(self, batch):
            return self.model(**batch).last_hidden_state[:, 0, :]

    # a smaller config for the merger fixture (it stores several full flat vectors)
    tiny2 = O.EncoderConfig(hidden=64, heads=1, layers=2, intermediate=32, vocab=40, max_pos=40)
    tiny2_sd = O.random_state_dict(O.roberta_param_shapes(tiny2), seed=1002, std=0.2)
    tiny2_sd = OrderedDict(("model." + k, tiny2_sd["model." + k]) for k in hf_roberta(tiny2, tiny2_sd).state_dict().keys())
    tiny, tiny_sd = tiny2, tiny2_sd
    ids = ids % (tiny2.vocab - 3) + 3
    for b, n in enumerate(lens):
        ids[b, 0] = 0
        if n > 1:
            ids[b, n - 1] = 2
        ids[b, n:] = tiny.pad_id

    def make_wrapper():
        return Wrapper(hf_roberta(tiny, tiny_sd))

    N = 3
    fts = [O.perturbed_state_dict(tiny_sd, seed=2000 + i, std=0.05) for i in range(N)]
    for ft in fts:  # reference checkpoints carry an extra key dropped by the key intersection
        ft["item_embeddings"] = torch.zeros(4, tiny.hidden)
    # shuffle one fine-tuned dict's key order to exercise the re-ordering (_factory.py:60-66)
    keys = list(fts[1].keys())
    fts[1] = OrderedDict((k, fts[1][k]) for k in reversed(keys))

    g2 = dict(cfg=tiny.__dict__, pretrain=tiny_sd, finetunes=fts, cases=[], input_ids=ids, attention_mask=mask)
    batch = BatchEncoding({"input_ids": ids, "attention_mask": mask})

    for learn, softmax_on in [("TASK_WISE", False), ("TASK_WISE", True), ("LAYER_WISE", False), ("LAYER_WISE", True)]:
        w = make_wrapper()
        mm = load_merging_module(
            merge_type=MergeType.TASK_VECTOR, learn_type=LearnType[learn], model=w, pretrain_state_dict=w.state_dict(),
            finetune_state_dicts=[dict(ft) for ft in fts], ignore_keys=set(), disable_softmax=not softmax_on,
            initial_per_weight=0.3,
        )
        groups = list(mm.per_weights.keys())
        gg = torch.Generator().manual_seed(77)
        weights = {
            "global_weights": {k: [float(0.8 + 0.4 * torch.rand(1, generator=gg))] for k in groups},
            "global_biases": {k: [float(0.05 * torch.randn(1, generator=gg))] for k in groups},
            # one extra trailing value: load_weights_from_dict truncates per_weights to N (_base.py:72)
            "per_weights": {k: (0.1 + 0.5 * torch.rand(N + 1, generator=gg)).tolist() for k in groups},
        }
        init_sd = {k: v.detach().clone() for k, v in mm.get_state_dict().items()}
        mm.load_weights_from_dict(weights)
        merged_sd = {k: v.detach().clone() for k, v in mm.get_state_dict().items()}
        with torch.no_grad():
            cls = mm.forward(batch).clone()  # reference module re-merges then runs the HF model functionally
        g2["cases"].append(dict(
            learn_type=learn, use_softmax=softmax_on, groups=groups, weights=weights,
            serialized=mm.serialize_weights(), shape_keys=list(mm.shape_dict.keys()),
            shapes=[tuple(s) for s in mm.shape_dict.values()],
            init_merged_flat=torch.cat([v.reshape(-1) for v in init_sd.values()]) if not softmax_on else None,
            merged_flat=torch.cat([v.reshape(-1) for v in merged_sd.values()]), cls=cls,
        ))
        g2["base_flat_checksum"] = float(mm.base_model_tensor.detach().double().sum())
        g2["tv_flat"] = mm.task_vectors_tensor.detach().clone()
    # fixed-alpha ModelMerger paths (merger.py:46-93): 'task_vector' sequential accumulation and 'linear'
    pre_aligned = OrderedDict((k, v) for k, v in tiny_sd.items())
    fts_aligned = [OrderedDict((k, ft[k]) for k in tiny_sd.keys()) for ft in fts]
    mg = ModelMerger(models=fts_aligned, base_model=pre_aligned, align_key_order=False)
    g2["model_merger_task_vector"] = torch.cat([v.reshape(-1) for v in mg.merge("task_vector", [0.5, 0.25, 0.7]).values()])
    g2["model_merger_linear"] = torch.cat([v.reshape(-1) for v in mg.merge("linear", [0.2, 0.3, 0.5]).values()])
    # an int64 buffer in the dicts is promoted to fp32 by flatten (model_operations.py:60-63)
    with_buf = OrderedDict([("model.embeddings.position_ids", torch.arange(10).view(1, 10))] + list(tiny_sd.items()))
    mg2 = ModelMerger(models=[with_buf], base_model=with_buf, align_key_order=False)
    g2["flatten_with_int_buffer_head"] = mg2.base_model[:12].clone()
    g2["flatten_with_int_buffer_dtype"] = str(mg2.base_model.dtype)
    torch.save(g2, OUT / "g2_merger.pt")
    print("G2 ok", [c["learn_type"] for c in g2["cases"]])

    # ------------------------------------------------------------------ G4 Recformer (a9, a10, a12)
    rm = load_by_path("_ref_recformer_models", REF / "rec_retrieval/module/models/encoder/recformer/models.py")
    from transformers import LongformerConfig

    def ref_recformer(cfg: "O.EncoderConfig", sd):
        hc = LongformerConfig(
            attention_window=[2 * cfg.one_sided_window] * cfg.layers, vocab_size=cfg.vocab, hidden_size=cfg.hidden,
            num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, intermediate_size=cfg.intermediate,
            max_position_embeddings=cfg.max_pos, type_vocab_size=1, pad_token_id=cfg.pad_id, layer_norm_eps=cfg.ln_eps,
            hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
        )
        hc.token_type_size = cfg.token_type_size
        hc.max_item_embeddings = cfg.max_item_embeddings
        hc.pooler_type = "cls"
        model = rm.RecformerModel(hc).eval()
        model.load_state_dict({k[len("model."):]: v for k, v in sd.items()}, strict=True)
        return model

    def run_recformer(model, batch):
        """recformer/models.py:273-361 with the mask built per :326-330 semantics."""
        am = model._merge_to_attention_mask(batch["attention_mask"], batch["global_attention_mask"])
        padding_len, input_ids, am, tt, pos, ip, _ = model._pad_to_window_size(
            input_ids=batch["input_ids"], attention_mask=am, token_type_ids=batch["token_type_ids"], position_ids=None,
            item_position_ids=batch["item_position_ids"], inputs_embeds=None, pad_token_id=model.config.pad_token_id,
        )
        ext = (1.0 - am.to(torch.float32)) * torch.finfo(torch.float32).min
        emb = model.embeddings(input_ids=input_ids, position_ids=pos, item_position_ids=ip, token_type_ids=tt)
        enc = model.encoder(emb, attention_mask=ext, padding_len=padding_len, output_hidden_states=True, return_dict=True)
        return emb[:, : emb.shape[1] - padding_len], enc

    g4 = dict(cases=[])
    for ci, (w1, layers, Lr, lens_r) in enumerate([(4, 2, 21, [21, 9, 1, 14]), (32, 1, 150, [150, 70, 33]), (32, 2, 100, [100, 40])]):
        rc = O.EncoderConfig(hidden=128, heads=2, layers=layers, intermediate=128, vocab=100, max_pos=200,
                             token_type_size=4, max_item_embeddings=51, one_sided_window=w1)
        rsd = O.random_state_dict(O.recformer_param_shapes(rc), seed=3000 + ci, std=0.2)
        model = ref_recformer(rc, rsd)
        assert set("model." + k for k in model.state_dict().keys()) == set(rsd.keys()), "recformer key set mismatch"
        rsd = OrderedDict(("model." + k, rsd["model." + k]) for k in model.state_dict().keys())
        Br = len(lens_r)
        rid = torch.randint(3, rc.vocab, (Br, Lr), generator=g)
        ram = torch.zeros(Br, Lr, dtype=torch.int64)
        rgm = torch.zeros(Br, Lr, dtype=torch.int64)
        rtt = torch.full((Br, Lr), 3, dtype=torch.int64)  # collate pad value (recformer_utils.py:99)
        rip = torch.zeros(Br, Lr, dtype=torch.int64)  # collate pad value (recformer_utils.py:97)
        for b, n in enumerate(lens_r):
            ram[b, :n] = 1
            rgm[b, 0] = 1
            rid[b, 0] = 0
            rid[b, n:] = rc.pad_id
            rtt[b, 0] = 0
            rtt[b, 1:n] = torch.randint(1, 3, (n - 1,), generator=g)
            rip[b, 1:n] = torch.clamp(1 + torch.arange(n - 1) // 6, max=50)
        batch_r = dict(input_ids=rid, attention_mask=ram, global_attention_mask=rgm, token_type_ids=rtt, item_position_ids=rip)
        with torch.no_grad():
            emb, enc = run_recformer(model, batch_r)
        g4["cases"].append(dict(
            cfg=rc.__dict__, state_dict=rsd, batch=batch_r, emb=emb.clone(),
            hidden_states=[h.clone() for h in enc.hidden_states], cls=enc.last_hidden_state[:, 0].clone(),
        ))
    torch.save(g4, OUT / "g4_recformer.pt")
    print("G4 ok", g4["cases"][0]["cls"][0, :4])

    # ------------------------------------------------------------------ G6 task-vector pre-processing (8(f).1)
    from rec_retrieval.merger.algorithms.ties import get_ties_vectors
    from rec_retrieval.merger.algorithms.localize_and_stitch import get_localize_and_stitch_vectors
    from rec_retrieval.merger.algorithms.pcb import get_pcb_vectors

    g6 = dict(cases=[])
    for ci, (Pn, Nn, dens) in enumerate([(20000, 3, 0.2), (5000, 5, 0.05), (4096, 2, 0.5)]):
        gg = torch.Generator().manual_seed(600 + ci)
        base6 = torch.randn(Pn, generator=gg) * 0.02
        models6 = [base6 + torch.randn(Pn, generator=gg) * 1e-3 for _ in range(Nn)]
        models6[0][100:140] = base6[100:140]          # exact zeros in a task vector
        if ci == 0:
            models6[1][:50] = base6[:50] + 0.25       # a block of equal large magnitudes well inside the top-k (never at the threshold)
        g6["cases"].append(dict(
            base=base6, models=models6, density=dens,
            ties=get_ties_vectors(base_model=base6, models=models6, density=dens).clone(),
            lns=get_localize_and_stitch_vectors(base_model=base6, models=models6, density=dens).clone(),
            pcb=get_pcb_vectors(base_model=base6, models=models6, density=dens).clone(),
        ))
    torch.save(g6, OUT / "g6_taskvector_algos.pt")
    print("G6 ok", g6["cases"][0]["ties"].abs().sum().item())

    for f in sorted(OUT.glob("*.pt")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
