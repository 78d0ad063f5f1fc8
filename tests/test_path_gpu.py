"""GPU parity of the whole path behind the reference's own API surface (merger -> encoder -> scoring ->
evaluator), against the golden vectors produced by the reference and against the CPU oracle."""
from collections import OrderedDict
from types import SimpleNamespace

import pytest
import torch

from oracle import c_oracle as CO
from oracle import ref_cpu as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _spec(cfgd, kind="roberta"):
    from mergerec_amd.engine import EncoderSpec

    return EncoderSpec(kind=kind, hidden=cfgd["hidden"], heads=cfgd["heads"], layers=cfgd["layers"], intermediate=cfgd["intermediate"],
                       vocab=cfgd["vocab"], max_pos=cfgd["max_pos"], pad_id=cfgd["pad_id"], ln_eps=cfgd["ln_eps"],
                       token_type_size=cfgd["token_type_size"], max_item_embeddings=cfgd["max_item_embeddings"],
                       one_sided_window=cfgd["one_sided_window"] if kind == "recformer" else -1)


def _dev_weights(sd, mode="bf16x6"):
    from mergerec_amd.engine import ArenaLayout, WeightSet

    views = OrderedDict((k, v.to(torch.float32)) for k, v in sd.items())
    layout = ArenaLayout(OrderedDict((k, tuple(v.shape)) for k, v in views.items()))
    return WeightSet(layout, layout.pack(views, DEV), mode).refresh()


MODES = ["f32", "bf16x6", "f16x3", "bf16x3"]
# hidden-state tolerance per GEMM arithmetic on the STRESS goldens (tiny models with std-0.2 weights: |activations| and attention
# logits far larger than in the real models); the contract of the path is 1e-4 on the cosine logits, asserted separately below
HID_TOL = {"f32": 1e-4, "bf16x6": 1e-4, "f16x3": 1e-4, "bf16x3": 1e-3}


def _cosine_logits_close(a, b, atol=1e-4):
    a, b = O.maybe_normalize(a), O.maybe_normalize(b)
    return float((a @ a.T - b @ b.T).abs().max()) <= atol and float((a - b).abs().max()) <= atol


def _unpack(hidden_packed, mask):
    B, L = mask.shape
    out = torch.zeros(B, L, hidden_packed.shape[1])
    out[mask.bool()] = hidden_packed.cpu()
    return out


# ------------------------------------------------------------------ encoder vs reference goldens
@pytest.mark.parametrize("mode", MODES)
def test_roberta_encoder_matches_library_golden(mode):
    from mergerec_amd.engine import EncoderRunner

    g3 = load_golden("g3_roberta.pt")
    run = EncoderRunner(_spec(g3["cfg"]))
    w = _dev_weights(g3["state_dict"], mode)
    batch = {"input_ids": g3["input_ids"], "attention_mask": g3["attention_mask"]}
    pb = run.pack(batch, DEV)
    cls, hidden = run.forward_packed(w, pb, normalize=False, return_hidden=True)
    m = g3["attention_mask"].bool()
    for li, (h, ref) in enumerate(zip(hidden, g3["hidden_states"])):
        got = _unpack(h, g3["attention_mask"])
        assert torch.allclose(got[m], ref[m], atol=HID_TOL[mode], rtol=1e-5), (li, (got[m] - ref[m]).abs().max())
    assert torch.allclose(cls.cpu(), g3["cls"], atol=HID_TOL[mode], rtol=1e-5)
    assert _cosine_logits_close(cls.cpu(), g3["cls"])
    fast = run.forward_packed(w, pb, normalize=False)  # last layer on CLS rows only
    assert torch.allclose(fast.cpu(), g3["cls"], atol=HID_TOL[mode], rtol=1e-5)
    nrm = run.forward_packed(w, pb, normalize=True).cpu()
    assert torch.allclose(nrm, O.maybe_normalize(g3["cls"]), atol=1e-5)


@pytest.mark.parametrize("mode", MODES)
def test_roberta_true_dims_layer_matches_library_golden(mode):
    from mergerec_amd.engine import EncoderRunner

    big = load_golden("g3_roberta.pt")["big"]
    cfg = O.EncoderConfig(**big["cfg"])
    sd = O.random_state_dict(O.roberta_param_shapes(cfg), seed=big["seed"], std=big["std"])
    run = EncoderRunner(_spec(big["cfg"]))
    pb = run.pack({"input_ids": big["input_ids"], "attention_mask": big["attention_mask"]}, DEV)
    _, hidden = run.forward_packed(_dev_weights(sd, mode), pb, normalize=False, return_hidden=True)
    m = big["attention_mask"].bool()
    assert torch.allclose(_unpack(hidden[0], big["attention_mask"])[m], big["emb"][m], atol=1e-5)
    got = _unpack(hidden[-1], big["attention_mask"])[m]
    assert torch.allclose(got, big["last"][m], atol=1e-4, rtol=1e-5), (got - big["last"][m]).abs().max()  # true dims: every mode within 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_recformer_encoder_matches_reference_golden(mode):
    from mergerec_amd.engine import EncoderRunner

    for case in load_golden("g4_recformer.pt")["cases"]:
        run = EncoderRunner(_spec(case["cfg"], "recformer"))
        w = _dev_weights({k: v for k, v in case["state_dict"].items()}, mode)
        b = case["batch"]
        pb = run.pack(b, DEV)
        cls, hidden = run.forward_packed(w, pb, normalize=False, return_hidden=True)
        m = b["attention_mask"].bool()
        for li, (h, ref) in enumerate(zip(hidden, case["hidden_states"])):
            got = _unpack(h, b["attention_mask"])
            assert torch.allclose(got[m], ref[m], atol=HID_TOL[mode], rtol=1e-5), (li, (got[m] - ref[m]).abs().max())
        assert torch.allclose(cls.cpu(), case["cls"], atol=HID_TOL[mode], rtol=1e-5)
        assert _cosine_logits_close(cls.cpu(), case["cls"])
        fast = run.forward_packed(w, pb, normalize=False)
        assert torch.allclose(fast.cpu(), case["cls"], atol=HID_TOL[mode], rtol=1e-5)


# ------------------------------------------------------------------ merger behind load_merging_module
NO_DROPOUT = {"hidden_dropout_prob": 0.0, "attention_probs_dropout_prob": 0.0}  # the HF configs the golden generators used (oracle/gen_golden*.py)


def _tiny_model(cfgd, kind="BLAIR_BASE", **extra):
    """dropout off by default: every fixture was recorded from the reference with hidden / attention dropout 0 (pass the two keys to turn
    it on for the training-graph dropout tests)"""
    from mergerec_amd.module import ModelType

    over = dict(hidden=cfgd["hidden"], heads=cfgd["heads"], layers=cfgd["layers"], intermediate=cfgd["intermediate"], vocab=cfgd["vocab"],
                max_pos=cfgd["max_pos"])
    return ModelType[kind].value(model_kwargs={"init_seed": 1, "spec_overrides": over, "device": DEV, **NO_DROPOUT, **extra})


def test_load_merging_module_matches_reference_golden():
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module

    g2 = load_golden("g2_merger.pt")
    batch = {"input_ids": g2["input_ids"], "attention_mask": g2["attention_mask"]}
    for case in g2["cases"]:
        model = _tiny_model(g2["cfg"])
        mm = load_merging_module(MergeType.TASK_VECTOR, LearnType[case["learn_type"]], model, g2["pretrain"],
                                 [dict(ft) for ft in g2["finetunes"]], set(), disable_softmax=not case["use_softmax"],
                                 initial_per_weight=0.3)
        assert list(mm.shape_dict.keys()) == case["shape_keys"]
        assert [tuple(s) for s in mm.shape_dict.values()] == case["shapes"]
        assert list(mm.per_weights.keys()) == case["groups"]
        assert torch.equal(mm.compact_task_vectors().cpu(), g2["tv_flat"])
        assert float(mm.compact_base().cpu().double().sum()) == g2["base_flat_checksum"]
        if case["init_merged_flat"] is not None:
            init = torch.cat([v.reshape(-1) for v in mm.get_state_dict().values()]).cpu()
            assert torch.equal(init, case["init_merged_flat"])
        mm.load_weights_from_dict(case["weights"])
        ser = mm.serialize_weights()
        for part in ("global_weights", "global_biases", "per_weights"):
            for k, v in case["serialized"][part].items():
                assert ser[part][k] == v
        sd = mm.get_state_dict()
        assert list(sd.keys()) == case["shape_keys"]
        merged = torch.cat([v.reshape(-1) for v in sd.values()]).cpu()
        if case["use_softmax"]:  # softmax over N alphas runs on the device: allow the last ulp of expf
            assert torch.allclose(merged, case["merged_flat"], atol=1e-7, rtol=1e-6)
        else:
            assert torch.equal(merged, case["merged_flat"]), case["learn_type"]
        cls = mm.forward(batch)  # re-merge into the arena, then the encoder reads it in place
        assert torch.allclose(cls.cpu(), case["cls"], atol=1e-4, rtol=1e-5), (cls.cpu() - case["cls"]).abs().max()


def test_model_merger_fixed_weight_merges_bit_exact():
    """ModelMerger(models, base).merge("task_vector" | "linear", weights) == the reference's own ModelMerger output (golden g2), bit for bit;
    argument checks as merger.py:46-93"""
    from mergerec_amd.merger import ModelMerger

    g2 = load_golden("g2_merger.pt")
    pre = g2["pretrain"]
    fts = [OrderedDict((k, ft[k]) for k in pre) for ft in g2["finetunes"]]
    mg = ModelMerger(models=fts, base_model=pre, align_key_order=False, device=DEV)
    flat = lambda sd: torch.cat([sd[k].reshape(-1) for k in pre]).cpu()
    tv = mg.merge("task_vector", [0.5, 0.25, 0.7])
    assert list(tv.keys()) == list(pre.keys()) and all(tv[k].shape == pre[k].shape for k in pre)
    assert torch.equal(flat(tv), g2["model_merger_task_vector"])
    assert torch.equal(flat(mg.merge("linear", [0.2, 0.3, 0.5])), g2["model_merger_linear"])
    # a single float is broadcast; sorted key order when asked to align; shuffled dicts are aligned, not rejected
    shuffled = [OrderedDict(reversed(list(ft.items()))) for ft in fts]
    mg2 = ModelMerger(models=shuffled, base_model=pre, align_key_order=True, device=DEV)
    assert list(mg2.shape_dict.keys()) == sorted(pre.keys())
    got = mg2.merge("task_vector", 0.25)
    want = O.merge_running(O.flatten_model(pre)[0], [O.flatten_model(ft)[0] for ft in fts], [0.25] * 3)
    assert torch.equal(flat(got), want)
    with pytest.raises(AssertionError, match="not aligned"):
        ModelMerger(models=shuffled, base_model=pre, align_key_order=False, device=DEV)
    with pytest.raises(ValueError, match="float or a list of floats"):
        mg.merge("linear", [1, 2, 3])
    with pytest.raises(ValueError, match="not supported"):
        mg.merge("slerp", 0.5)
    with pytest.raises(ValueError, match="requires a base model"):
        ModelMerger(models=fts, device=DEV).merge("task_vector", 0.5)
    head = ModelMerger(models=fts, device=DEV)  # no base: all models are merged (the reference keeps the first as the head of the list)
    assert torch.equal(flat(head.merge("linear", [0.2, 0.3, 0.5])), g2["model_merger_linear"])


def test_model_merger_ties_pcb_dare_match_reference():
    """ModelMerger.merge("ties" | "pcb" | "dare", weights, density=...) against the reference's own ModelMerger on the g2 models (fixture g21,
    oracle/gen_golden_model_merger.py): ties (trim, sum in model order) and dare (torch's dropout draws from the global CPU generator, seeded
    as the reference was) bit for bit; pcb to the tolerance of its exp / tanh chain."""
    from mergerec_amd.merger import ModelMerger

    g2, g21 = load_golden("g2_merger.pt"), load_golden("g21_model_merger.pt")
    pre = g2["pretrain"]
    fts = [OrderedDict((k, ft[k]) for k in pre) for ft in g2["finetunes"]]
    mg = ModelMerger(models=fts, base_model=pre, align_key_order=False, device=DEV)
    flat = lambda sd: torch.cat([sd[k].reshape(-1).float() for k in pre]).cpu()
    w = list(g21["weights"])
    for name in ("ties", "ties_dense"):
        c = g21["cases"][name]
        got = flat(mg.merge("ties", w, **c["kwargs"]))
        assert torch.equal(got, c["merged_flat"]), (name, (got - c["merged_flat"]).abs().max())
    c = g21["cases"]["dare"]
    torch.manual_seed(c["seed"])
    got = flat(mg.merge("dare", w, **c["kwargs"]))
    assert torch.equal(got, c["merged_flat"]), (got - c["merged_flat"]).abs().max()
    c = g21["cases"]["pcb"]
    got = flat(mg.merge("pcb", w, **c["kwargs"]))
    assert torch.allclose(got, c["merged_flat"], rtol=2e-4, atol=1e-7), (got - c["merged_flat"]).abs().max()
    for mt in ("ties", "dare", "pcb"):
        with pytest.raises(ValueError, match="requires a base model"):
            ModelMerger(models=fts, device=DEV).merge(mt, 0.5, density=0.2)


def test_merging_module_errors_mirror_reference():
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module

    g2 = load_golden("g2_merger.pt")
    model = _tiny_model(g2["cfg"])
    with pytest.raises(AssertionError):
        load_merging_module("TASK_VECTOR", LearnType.TASK_WISE, model, g2["pretrain"], g2["finetunes"], set())
    with pytest.raises(ValueError):
        load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, model, [1, 2], g2["finetunes"], set())
    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, model, g2["pretrain"], [dict(f) for f in g2["finetunes"]], set())
    with pytest.raises(AssertionError):
        mm.load_weights_from_dict({"global_weights": {"nope": [1.0]}, "global_biases": {}, "per_weights": {}})
    with pytest.raises(AssertionError):
        load_merging_module(MergeType.TIES, LearnType.TASK_WISE, _tiny_model(g2["cfg"]), g2["pretrain"], [dict(f) for f in g2["finetunes"]], set())


# ------------------------------------------------------------------ the whole path vs the oracle
@pytest.mark.parametrize("kind", ["BLAIR_BASE", "RECFORMER_BASE"])
def test_end_to_end_merge_encode_score_evaluate(kind):
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.synthetic import make_domain
    from mergerec_amd.utils import test_model_on_dataloaders as test_model

    rec = kind.startswith("RECFORMER")
    over = dict(hidden=128, heads=2, layers=2, intermediate=256, vocab=300, max_pos=200)
    mk = {"init_seed": 11, "spec_overrides": over, "device": DEV}
    model = ModelType[kind].value(model_kwargs=dict(mk))
    pre = OrderedDict((k, v.cpu().clone()) for k, v in model.state_dict().items())
    fts = [O.perturbed_state_dict(pre, seed=50 + i, std=0.02) for i in range(3)]
    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType.LAYER_WISE if rec else LearnType.TASK_WISE, model, pre, fts, set(),
                             disable_softmax=True)
    groups = list(mm.per_weights.keys())
    gg = torch.Generator().manual_seed(3)
    weights = {"global_weights": {k: [1.0] for k in groups}, "global_biases": {k: [0.0] for k in groups},
               "per_weights": {k: (0.1 + 0.5 * torch.rand(3, generator=gg)).tolist() for k in groups}}
    mm.load_weights_from_dict(weights)
    state_dict = {k: v.detach() for k, v in mm.get_state_dict().items()}
    model2 = ModelType[kind].value(model_kwargs=dict(mk))
    model2.load_state_dict(state_dict)  # merge_test.py:71-80
    module = RecModule(model=model2, evaluator=Evaluator(["NDCG", "RECALL"], [1, 5, 10, 50]), similarity="cosine")
    dom = make_domain("Toy", n_items=333, n_users=150, batch_size=32, vocab=over["vocab"], seed=9,
                      kind="recformer" if rec else "roberta", max_seq_len=160, item_len_scale=0.3)
    metric_dict, metrics, scores, labels = test_model(module, [dom.item_batches], [dom.sequence_batches], ["Toy"])
    # the (users, items) matrix is always available, as upstream (utils.py:113, module.py:344-352): produced on first access here
    assert len(scores) == 1 and scores[0].shape == (dom.labels.numel(), module.item_embeddings.shape[0])
    assert torch.equal(scores[0], module.eval_scores) and not module.keep_scores
    assert torch.equal(scores[0], CO.gemm_nt(module.eval_user_embeddings, module.item_embeddings.detach().cpu()))   # the bits the kernel ranked
    assert torch.equal(torch.gather(scores[0], 1, module.eval_topk_indices[:, :1]).squeeze(1), scores[0].max(1).values)

    # ---- oracle: same arithmetic on CPU
    base, shape_dict = O.flatten_model(pre)
    tv = O.get_task_vectors(base, [O.flatten_model(OrderedDict((k, ft[k]) for k in pre))[0] for ft in fts])
    if rec:
        grp = O.group_parameters_by_layer(shape_dict)
        alpha = {k: torch.tensor(weights["per_weights"][k]) for k in grp}
        merged = O.merge_layer_wise(base, tv, grp, alpha)
    else:
        merged = O.merge_task_wise(base, tv, torch.tensor(weights["per_weights"]["all"]))
    got_merged = torch.cat([v.reshape(-1) for v in state_dict.values()]).cpu()
    assert torch.equal(got_merged, merged), "merged parameters must be bit-exact"
    sd = O.get_state_dict(merged, shape_dict)
    cfg = O.EncoderConfig(hidden=128, heads=2, layers=2, intermediate=256, vocab=300, max_pos=200,
                          token_type_size=4 if rec else 1, max_item_embeddings=51 if rec else 0, one_sided_window=32 if rec else 0)

    def enc(b):
        if rec:
            return O.recformer_encode(sd, b["input_ids"], b["attention_mask"], b["global_attention_mask"], b["token_type_ids"],
                                      b["item_position_ids"], cfg, "model.")
        return O.roberta_encode(sd, b["input_ids"], b["attention_mask"], cfg, "model.")

    E = O.maybe_normalize(torch.cat([enc(b.items) for b in dom.item_batches]))
    U = O.maybe_normalize(torch.cat([enc(b.sequence) for b in dom.sequence_batches]))
    assert torch.allclose(module.item_embeddings.detach().cpu(), E, atol=1e-4)
    assert torch.allclose(module.eval_user_embeddings, U, atol=1e-4)
    ref_scores = O.score(U, E)
    # exact vs the oracle evaluator on the FMA-chain scores of the GPU embeddings; tie-aware vs the pure-CPU pipeline
    chain = CO.gemm_nt(module.eval_user_embeddings, module.item_embeddings.detach().cpu())
    exact = O.evaluate(chain, dom.labels, ["NDCG", "RECALL"], [1, 5, 10, 50], "test/")
    for k, v in exact.items():
        assert metrics[0][k] == v, (k, metrics[0][k], v)
    want = O.evaluate(ref_scores, dom.labels, ["NDCG", "RECALL"], [1, 5, 10, 50], "test/")
    for k, v in want.items():
        assert abs(metrics[0][k] - v) <= 0.02, (k, metrics[0][k], v)   # near-tie swaps only (150 users)
    assert abs(metrics[0]["test/loss"] - O.ce_loss(ref_scores, dom.labels, 0.05)) < 2e-3
    _, oi = O.topk_canonical(ref_scores, 50)
    assert O.ranks_equal_up_to_ties(ref_scores, module.eval_topk_indices, oi, atol=2e-4)
    assert list(metric_dict.keys())[0].startswith("test/dataset_0/")
    assert torch.equal(labels[0], dom.labels)


# ------------------------------------------------------------------ CLI drop-in (merge_test.py surface)
def test_merge_test_cli_synthetic(tmp_path):
    import csv
    import sys

    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
    import merge_test

    out = tmp_path / "metrics.csv"
    over = ["--model_type", "blair_base", "--model_kwargs", "init_seed", "7", "--finetune_checkpoint_paths", "synthetic:1", "synthetic:2",
            "--merge_type", "task_vector", "--learn_type", "task_wise", "--weight_file", "uniform", "--weight_file_line", "0.5",
            "--data_paths", "synthetic:Tiny:300:96", "--batch_size", "32", "--metrics_path", str(out), "--lora.enable", "False"]
    # tiny architecture through spec_overrides is not a CLI flag: patch the spec factory for the test
    from mergerec_amd.engine import EncoderSpec
    from mergerec_amd.module import models

    old = models.BLaIRBase.SPEC
    models.BLaIRBase.SPEC = staticmethod(lambda: EncoderSpec(hidden=128, heads=2, layers=2, intermediate=256, vocab=50265, max_pos=514))
    try:
        metrics = merge_test.main(over)
    finally:
        models.BLaIRBase.SPEC = staticmethod(old)
    assert set(metrics[0]) >= {"test/NDCG@10", "test/Recall@50", "test/loss"}
    rows = list(csv.DictReader(open(out)))
    assert rows[0]["dataset"] == "Tiny" and float(rows[0]["test/Recall@50"]) == metrics[0]["test/Recall@50"]


def test_merge_test_cli_json_dataset_with_local_tokenizer(tmp_path):
    """dataset directory in the reference's JSON format + a local tokenizer directory -> datamodule route -> metrics, and the
    same numbers as feeding the datamodule's batches to the oracle encoder + evaluator on the CPU"""
    import sys

    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
    import merge_test
    from mergerec_amd.engine import EncoderSpec
    from mergerec_amd.module import models
    from tests.conftest import GOLDEN

    emb = tmp_path / "user.pt"
    over = ["--model_type", "blair_base", "--model_kwargs", "init_seed", "7", "--finetune_checkpoint_paths", "synthetic:1", "synthetic:2",
            "--merge_type", "task_vector", "--learn_type", "task_wise", "--weight_file", "average", "--data_paths", str(GOLDEN / "mini_dataset"),
            "--tokenizer_path", str(GOLDEN / "mini_tokenizer"), "--batch_size", "8", "--max_seq_len", "96", "--max_attribute_len", "12",
            "--max_items", "20", "--user_embeddings_path", str(emb)]
    old = models.BLaIRBase.SPEC
    models.BLaIRBase.SPEC = staticmethod(lambda: EncoderSpec(hidden=128, heads=2, layers=2, intermediate=256, vocab=50265, max_pos=514))
    try:
        metrics = merge_test.main(over)
    finally:
        models.BLaIRBase.SPEC = staticmethod(old)
    assert set(metrics[0]) >= {"test/NDCG@10", "test/Recall@50", "test/loss"}
    users = torch.load(emb)[0]
    assert users.shape == (40, 128) and torch.allclose(users.norm(dim=-1), torch.ones(40), atol=1e-5)


def test_precision_flag_selects_arithmetic():
    """Lightning precision strings: 32-true keeps the model's mode, the reference's default bf16-mixed runs f16x3, junk raises"""
    from mergerec_amd.module import RecModule
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.data import load_domain
    from mergerec_amd.utils import Trainer, precision_to_gemm_mode, test_model_on_dataloaders as test_model

    assert precision_to_gemm_mode("32-true") is None and precision_to_gemm_mode("bf16-mixed") == "f16x3"
    with pytest.raises(ValueError):
        precision_to_gemm_mode("int4")
    model = _tiny_model(dict(hidden=128, heads=2, layers=2, intermediate=256, vocab=1000, max_pos=514))
    dom = load_domain("synthetic:Tiny:120:40", vocab=1000)
    mod = RecModule(model=model, evaluator=Evaluator(metrics=["NDCG", "RECALL"], ks=[10]), negative_sample=None, similarity="cosine")
    _, m32, _, _ = test_model(mod, [dom.item_dataloader(16)], [dom.sequence_dataloader(16)], ["Tiny"], precision="32-true")
    assert model._weights.mode == "bf16x6"
    _, m16, _, _ = test_model(mod, [dom.item_dataloader(16)], [dom.sequence_dataloader(16)], ["Tiny"], precision="bf16-mixed")
    assert model._weights.mode == "f16x3"
    assert abs(m32[0]["test/loss"] - m16[0]["test/loss"]) < 1e-3 and m32[0].keys() == m16[0].keys()


def test_pooling_methods():
    """encoder/_base.py:41-49: 'cls' (default) = last_hidden_state[:, 0]; 'pooler' = RobertaPooler's tanh(dense(h_cls)) on the library's own
    CLS rows of fixture g3; 'mean' = transformers' last_hidden_state.mean(dim=1) over the padded width, pad positions included (g3 stores
    the padded hidden states transformers produced), for every arithmetic, also when batches of different widths are coalesced; Recformer
    refuses 'mean' with the reason; junk raises upstream's ValueError."""
    from mergerec_amd.module import ModelType
    from tests.conftest import load_golden

    g3 = load_golden("g3_roberta.pt")
    c = g3["cfg"]
    over = dict(hidden=c["hidden"], heads=c["heads"], layers=c["layers"], intermediate=c["intermediate"], vocab=c["vocab"], max_pos=c["max_pos"])
    batch = {"input_ids": g3["input_ids"].to(DEV), "attention_mask": g3["attention_mask"].to(DEV)}
    sd = g3["state_dict"]
    for method in ("cls", "pooler"):
        model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 0, "spec_overrides": over, "device": DEV, "gemm_mode": "f32"}, pooling_method=method)
        model.load_state_dict(sd)
        out = model(batch).cpu()
        want = g3["cls"] if method == "cls" else torch.tanh(torch.nn.functional.linear(g3["cls"], sd["model.pooler.dense.weight"], sd["model.pooler.dense.bias"]))
        assert float((out - want).abs().max()) < 1e-4, method
        norm = model.encode_normalized(batch, True).cpu()
        assert torch.allclose(norm, torch.nn.functional.normalize(want, dim=-1), atol=1e-5)
    want = g3["hidden_states"][-1].mean(dim=1)  # transformers, (B, L, d) with the pad positions in it
    for mode, tol in (("f32", 2e-5), ("bf16x6", 2e-5), ("f16x3", 5e-5)):
        model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 0, "spec_overrides": over, "device": DEV, "gemm_mode": mode}, pooling_method="mean")
        model.load_state_dict(sd)
        out = model(batch).cpu()
        assert float((out - want).abs().max()) < tol * max(1.0, float(want.abs().max())), (mode, float((out - want).abs().max()))
        norm = model.encode_normalized(batch, True).cpu()
        assert torch.allclose(norm, torch.nn.functional.normalize(want, dim=-1), atol=2e-5), mode
    # rows keep the padded width of the batch they came in when batches are coalesced (data.coalesce_batches): the first three rows cut to
    # their own batch width 9 and the whole batch at 23, merged into one kernel pass, must give what the two separate calls give
    from mergerec_amd.data import coalesce_batches
    from mergerec_amd.model_batch import BatchItem

    ids, mask = g3["input_ids"], g3["attention_mask"]
    narrow = {"input_ids": ids[1:3, :9].clone(), "attention_mask": mask[1:3, :9].clone()}  # lengths 1 and 7 inside width 9
    wide = {"input_ids": ids, "attention_mask": mask}
    sep = torch.cat([model(BatchItem(items=narrow).to(DEV).items).cpu(), model(BatchItem(items=wide).to(DEV).items).cpu()])
    merged = list(coalesce_batches([BatchItem(items=narrow), BatchItem(items=wide)], max_tokens=10 ** 6))
    assert len(merged) == 1 and merged[0].items["input_ids"].shape == (7, 23)
    together = model(merged[0].to(DEV).items).cpu()
    assert torch.allclose(together, sep, atol=1e-6), float((together - sep).abs().max())
    assert float((together[:2] - together[3:5]).abs().max()) > 1e-3  # same sequences, different padded widths: different means, as upstream
    with pytest.raises(NotImplementedError, match="RoBERTa"):
        ModelType.RECFORMER_BASE.value(model_kwargs={"init_seed": 0, "device": DEV}, pooling_method="mean")
    with pytest.raises(ValueError, match="Invalid pooling method"):
        ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 0, "spec_overrides": over, "device": DEV}, pooling_method="max")
    with pytest.raises(RuntimeError, match="pooler head"):
        rec = ModelType.RECFORMER_BASE.value(model_kwargs={"init_seed": 0, "device": DEV, "spec_overrides": dict(hidden=128, heads=2, layers=1, intermediate=128, vocab=300, max_pos=200)},
                                             pooling_method="pooler")
        ids = torch.tensor([[0, 5, 6, 2]], device=DEV)
        rec({"input_ids": ids, "attention_mask": torch.ones_like(ids), "global_attention_mask": torch.tensor([[1, 0, 0, 0]], device=DEV),
             "token_type_ids": torch.tensor([[0, 1, 2, 2]], device=DEV), "item_position_ids": torch.tensor([[0, 1, 1, 1]], device=DEV)})


def test_lazy_eval_scores_are_the_epochs_bits_even_if_the_catalog_is_rewritten_afterwards():
    """module.py:344-352: ``eval_scores`` is what the epoch computed.  Here it is produced on first access -- from a snapshot of the table
    the kernel ranked, not from the live ``item_embeddings``, which a catalog refresh or the next domain's encode may rewrite in place."""
    from oracle import c_oracle as CO
    from mergerec_amd.data import load_domain
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.module import RecModule
    from mergerec_amd.utils import test_model_on_dataloaders as test_model

    model = _tiny_model(dict(hidden=128, heads=2, layers=2, intermediate=256, vocab=1000, max_pos=514))
    dom = load_domain("synthetic:Tiny:120:40", vocab=1000)
    mod = RecModule(model=model, evaluator=Evaluator(metrics=["NDCG", "RECALL"], ks=[10]), negative_sample=None, similarity="cosine")
    test_model(mod, [dom.item_dataloader(16)], [dom.sequence_dataloader(16)], ["Tiny"])
    assert mod._eval_scores is None and mod._eval_scores_lazy is not None
    ranked = mod.item_embeddings.detach().cpu().clone()
    mod.item_embeddings.data[:60] = 0.25                          # in place, after the epoch, before anyone read eval_scores
    scores = mod.eval_scores
    assert torch.equal(scores, CO.gemm_nt(mod.eval_user_embeddings, ranked))
    assert torch.equal(torch.gather(scores, 1, mod.eval_topk_indices[:, :1]).squeeze(1), scores.max(1).values)
    assert mod._eval_scores_lazy is None and mod.eval_scores is scores  # cached, snapshot released


def test_pipelined_merges_are_bit_identical_and_follow_alpha():
    """``pipeline_merges``: the merge of step s + 1 runs on a second stream into a second arena while step s encodes.  Same bits as the
    in-stream merge on every step; a coefficient change between steps (in place, by assignment, through load_weights_from_dict) is never
    served from the speculative arena."""
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.module import ModelType
    from mergerec_amd.synthetic import make_domain

    over = dict(hidden=128, heads=2, layers=2, intermediate=256, vocab=300, max_pos=200)
    dom = make_domain("Toy", n_items=64, n_users=8, batch_size=32, vocab=300, seed=5, max_seq_len=120, item_len_scale=0.3)
    batches = [b.items for b in dom.item_batches]

    def run(pipelined):
        model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 21, "spec_overrides": over, "device": DEV})
        pre = OrderedDict((k, v.cpu().clone()) for k, v in model.state_dict().items())
        fts = [O.perturbed_state_dict(pre, seed=90 + i, std=0.05) for i in range(3)]
        mm = load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, model, pre, fts, set(), disable_softmax=True)
        if pipelined:
            mm.pipeline_merges(True)
        outs = []
        for step in range(8):
            if step == 3:    # in-place update, as an optimizer step does it
                with torch.no_grad():
                    mm.per_weights["all"].mul_(0.5)
            if step == 5:    # the alpha file path
                mm.load_weights_from_dict({"global_weights": {"all": [0.9]}, "global_biases": {"all": [0.01]}, "per_weights": {"all": [0.3, 0.1, 0.2]}})
            mm.load_weights(force=True)
            outs.append(model.encode_normalized(batches[step % len(batches)], True).clone())
            outs.append(torch.cat([v.reshape(-1) for v in model.state_dict().values()]).clone())
        torch.cuda.synchronize()
        return outs

    a, b = run(False), run(True)
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), i


def test_extract_and_finetune_test_single_model(tmp_path):
    """Lightning-style checkpoint -> scripts/extract.py -> finetune_test.py (single-model path): state_dict.pt keys are
    ``model.model.*`` + ``item_embeddings``; the CLI must load it (prefix stripped, item_embeddings dropped) and evaluate"""
    import sys

    root = __import__("pathlib").Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    sys.path.insert(0, str(root / "scripts"))
    import extract
    import finetune_test
    from mergerec_amd.engine import EncoderSpec
    from mergerec_amd.module import models
    from tests.conftest import GOLDEN

    old = models.BLaIRBase.SPEC
    models.BLaIRBase.SPEC = staticmethod(lambda: EncoderSpec(hidden=128, heads=2, layers=2, intermediate=256, vocab=50265, max_pos=514))
    try:
        m = models.BLaIRBase(model_kwargs={"init_seed": 3})
        sd = {"model." + k: v.cpu().clone() + 0.01 for k, v in m.state_dict().items()}  # RecModule.model.<encoder keys "model.*">
        sd["item_embeddings"] = torch.randn(60, 128)
        torch.save({"state_dict": sd, "epoch": 3}, tmp_path / "last.ckpt")
        extract.extract_checkpoint(tmp_path / "last.ckpt", tmp_path / "out")
        assert torch.equal(torch.load(tmp_path / "out" / "item_embedding.pt"), sd["item_embeddings"])
        args = ["--model_type", "blair_base", "--model_kwargs", "init_seed", "3", "--finetune_checkpoint_path", str(tmp_path / "out" / "state_dict.pt"),
                "--data_path", str(GOLDEN / "mini_dataset"), "--tokenizer_path", str(GOLDEN / "mini_tokenizer"), "--data_split", "val",
                "--batch_size", "8", "--max_seq_len", "96", "--max_attribute_len", "12", "--max_items", "20",
                "--item_embeddings_path", str(tmp_path / "items.pt")]
        metrics = finetune_test.main(args)
        base = finetune_test.main(args[:5] + ["--finetune_checkpoint_path", "synthetic:0"] + args[7:])
    finally:
        models.BLaIRBase.SPEC = staticmethod(old)
    assert set(metrics[0]) >= {"test/NDCG@10", "test/Recall@50", "test/loss"}
    assert torch.load(tmp_path / "items.pt")[0].shape == (60, 128)
    assert metrics[0]["test/loss"] != base[0]["test/loss"]  # the checkpoint (+0.01 on every weight) was really loaded


def test_merge_autograd_alpha_gradient():
    """a20: d(loss)/d(alpha params) through the HIP merge backward == torch autograd on the reference expression."""
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module

    g2 = load_golden("g2_merger.pt")
    for learn, softmax_on in (("TASK_WISE", False), ("LAYER_WISE", True)):
        model = _tiny_model(g2["cfg"])
        mm = load_merging_module(MergeType.TASK_VECTOR, LearnType[learn], model, g2["pretrain"], [dict(f) for f in g2["finetunes"]], set(),
                                 disable_softmax=not softmax_on, initial_per_weight=0.3)
        gen = torch.Generator().manual_seed(1)
        P = mm.layout.padded_numel
        probe = torch.randn(P, generator=gen).to(DEV)
        merged = mm.merged_params()
        loss = (merged * probe).sum()
        loss.backward()
        # reference expression on CPU with autograd
        base = mm.base_model_tensor.detach().cpu()
        tv = mm.task_vectors_tensor.detach().cpu()
        probe_c = probe.cpu()
        groups, seg_off, seg_gid = (["all"], None, [0]) if learn == "TASK_WISE" else mm.layout.group_segments()
        params = {k: (torch.ones(1, requires_grad=True), torch.zeros(1, requires_grad=True), torch.full((tv.shape[0],), 0.3, requires_grad=True)) for k in groups}
        total = 0.0
        bounds = [0, P] if seg_off is None else seg_off.tolist()
        for s, gid in enumerate(seg_gid):
            gw, gb, per = params[groups[gid]]
            pw = torch.softmax(per, 0) if softmax_on else per
            a = gw * pw + gb
            lo, hi = bounds[s], bounds[s + 1]
            m = base[lo:hi] + (a.unsqueeze(1) * tv[:, lo:hi]).sum(0)
            total = total + (m * probe_c[lo:hi]).sum()
        total.backward()
        for k in groups:
            gw, gb, per = params[k]
            scale = max(1.0, float(per.grad.abs().max()))
            assert torch.allclose(mm.per_weights[k].grad.cpu(), per.grad, rtol=1e-3, atol=1e-3 * scale), (k, mm.per_weights[k].grad, per.grad)
            assert torch.allclose(mm.global_weights[k].grad.cpu(), gw.grad, rtol=1e-3, atol=1e-3 * scale)
            assert torch.allclose(mm.global_biases[k].grad.cpu(), gb.grad, rtol=1e-3, atol=1e-3 * scale)


# ------------------------------------------------------------------ 8(f).1: TIES / Localize-and-Stitch pre-processing
def test_ties_and_lns_kernels_match_reference_golden():
    from mergerec_amd import ops

    for case in load_golden("g6_taskvector_algos.pt")["cases"]:
        base, models, dens = case["base"], case["models"], case["density"]
        P, N = base.numel(), len(models)
        k = int(dens * P)
        tv = torch.stack([m - base for m in models]).to(DEV)
        # top-k mask: exactly k kept, equal to the oracle's lowest-index tie rule
        y, m = ops.abs_topk_mask(tv[0].contiguous(), k, want_mask=True)
        want = O.topk_abs_mask(tv[0].cpu(), k)
        assert int(m.sum()) == k and torch.equal(m.cpu().bool(), want)
        sp = torch.stack([ops.abs_topk_mask(tv[i].contiguous(), k)[0] for i in range(N)])
        got = ops.ties_combine(sp).cpu()
        assert torch.equal(got, case["ties"]), (got - case["ties"]).abs().max()
        masks = torch.stack([ops.abs_topk_mask(tv[i].contiguous(), k, want_mask=True)[1] for i in range(N)])
        got = ops.lns_combine(tv.contiguous(), masks).cpu()
        assert torch.equal(got, case["lns"]), (got - case["lns"]).abs().max()


def test_topk_mask_ties_at_threshold_and_sizes():
    from mergerec_amd import ops

    g = torch.Generator().manual_seed(5)
    for n, k in [(5, 2), (2048, 2048), (70001, 14000), (4097, 1)]:
        x = torch.randn(n, generator=g)
        x[torch.randperm(n, generator=g)[: n // 3]] = 0.5  # a large block of exact ties that straddles the threshold
        x[::7] *= -1
        y, m = ops.abs_topk_mask(x.to(DEV), k, want_mask=True)
        want = O.topk_abs_mask(x, k)
        assert int(m.sum()) == k
        assert torch.equal(m.cpu().bool(), want)
        assert torch.equal(y.cpu(), torch.where(want, x, torch.zeros_like(x)))


@pytest.mark.parametrize("merge_type", ["TIES", "LOCALIZE_AND_STITCH"])
def test_load_merging_module_ties_lns(merge_type):
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module

    g2 = load_golden("g2_merger.pt")
    model = _tiny_model(g2["cfg"])
    fts = [dict(f) for f in g2["finetunes"]]
    mm = load_merging_module(MergeType[merge_type], LearnType.TASK_WISE, model, g2["pretrain"], fts, set(), ties_density=0.2,
                             disable_softmax=True, initial_per_weight=0.4)
    pre, al = O.align_state_dicts(g2["pretrain"], g2["finetunes"])
    base, _ = O.flatten_model(pre)
    models = [O.flatten_model(f)[0] for f in al]
    want_tv = O.ties_vectors(base, models, 0.2) if merge_type == "TIES" else O.localize_and_stitch_vectors(base, models, 0.2)
    assert torch.equal(mm.compact_task_vectors().cpu(), want_tv)
    merged = torch.cat([v.reshape(-1) for v in mm.get_state_dict().values()]).cpu()
    assert torch.equal(merged, O.merge_task_wise(base, want_tv, torch.full((3,), 0.4)))


def test_scoring_edge_cases_behave_like_the_reference():
    """The evaluation loop on degenerate shapes, and the two errors upstream raises from torch: ``torch.topk(scores, max(ks))`` on a catalog
    smaller than max(ks) (evaluator/evaluator.py:43: RuntimeError) and ``F.cross_entropy`` on a label outside the catalog (module.py:356:
    IndexError "Target ... is out of bounds.") -- the scoring kernel itself reads no memory for such a label."""
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.synthetic import make_domain
    from mergerec_amd.utils import test_model_on_dataloaders as test_model

    over = dict(hidden=128, heads=2, layers=1, intermediate=128, vocab=300, max_pos=200)

    def run(ks, n_items, n_users, bs=32, mutate=None, module=None):
        mod = module or RecModule(model=ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 3, "spec_overrides": over, "device": DEV}),
                                  evaluator=Evaluator(["NDCG", "RECALL"], ks), similarity="cosine")
        dom = make_domain("Toy", n_items=n_items, n_users=n_users, batch_size=bs, vocab=300, seed=5, max_seq_len=120, item_len_scale=0.3)
        if mutate:
            mutate(dom)
        return mod, dom, test_model(mod, [dom.item_batches], [dom.sequence_batches], ["Toy"])[1][0]

    with pytest.raises(RuntimeError, match="selected index k out of range"):
        run([1, 5, 10, 50], 30, 40)
    with pytest.raises(RuntimeError, match="selected index k out of range"):
        Evaluator(["NDCG"], [50])(torch.randn(4, 30, device=DEV), torch.zeros(4, dtype=torch.int64, device=DEV))
    _, _, m = run([1, 5, 10], 30, 40)  # k up to the catalog size is fine
    assert m["test/Recall@10"] >= m["test/Recall@5"] >= m["test/Recall@1"]
    _, _, m = run([1], 1, 5)  # a one-item catalog: every label is item 0, cosine scores, loss = log(1) = 0
    assert m == {"test/NDCG@1": 1.0, "test/Recall@1": 1.0, "test/loss": 0.0}
    _, dom1, one = run([1, 5], 64, 1)
    assert set(one) == {"test/NDCG@1", "test/NDCG@5", "test/Recall@1", "test/Recall@5", "test/loss"} and one["test/loss"] == one["test/loss"]
    _, _, a = run([1, 5], 64, 7, bs=1)  # one-row batches all the way through
    assert a["test/loss"] == a["test/loss"] and 0.0 <= a["test/Recall@5"] <= 1.0
    # metric keys keep the order of --ks as given (evaluator.py:12-15 builds them metric-major in that order)
    _, _, m = run([10, 5, 1], 64, 40)
    assert list(m) == ["test/NDCG@10", "test/NDCG@5", "test/NDCG@1", "test/Recall@10", "test/Recall@5", "test/Recall@1", "test/loss"]
    # a module evaluated twice starts from clean state
    mod, dom, first = run([1, 10], 64, 40)
    assert test_model(mod, [dom.item_batches], [dom.sequence_batches], ["Toy"])[1][0] == first

    def bad_label(value):
        def f(dom):
            dom.sequence_batches[0].labels = dom.sequence_batches[0].labels.clone()
            dom.sequence_batches[0].labels[0] = value
        return f

    with pytest.raises(IndexError, match="Target 10000 is out of bounds"):
        run([1, 10], 64, 40, mutate=bad_label(10_000))
    with pytest.raises(IndexError, match="Target -1 is out of bounds"):
        run([1, 10], 64, 40, mutate=bad_label(-1))


def test_coalesced_batches_give_identical_results():
    """Macro-batching of the dataloader stream (padding-invariant kernels) must not change any per-sequence result."""
    from mergerec_amd.data import coalesce_batches
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.synthetic import make_domain
    from mergerec_amd.utils import Trainer
    from mergerec_amd.module.callbacks import ItemEncodingCallback

    over = dict(hidden=128, heads=2, layers=2, intermediate=256, vocab=300, max_pos=200)
    model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 3, "spec_overrides": over, "device": DEV, "gemm_mode": "f32"})
    dom = make_domain("Toy", n_items=150, n_users=90, batch_size=16, vocab=300, seed=4, max_seq_len=150, item_len_scale=0.3)
    assert len(list(coalesce_batches(dom.sequence_batches, 2000))) < len(dom.sequence_batches)
    res = []
    for tokens in (0, 2000, 1 << 20):
        module = RecModule(model=model, evaluator=Evaluator(["NDCG", "RECALL"], [1, 5, 10, 50]), similarity="cosine")
        cb = ItemEncodingCallback(dom.item_batches)
        tr = Trainer(callbacks=[cb], coalesce_tokens=tokens)
        m = tr.test(module, dom.sequence_batches)[0]
        res.append((m, module.item_embeddings.detach().cpu().clone(), module.eval_user_embeddings.clone(), module.eval_topk_indices.clone()))
    for m, e, u, idx in res[1:]:
        assert m == res[0][0]
        assert torch.equal(e, res[0][1]) and torch.equal(u, res[0][2]) and torch.equal(idx, res[0][3])


def test_pcb_vectors_match_reference_golden():
    from mergerec_amd import ops

    for case in load_golden("g6_taskvector_algos.pt")["cases"]:
        base, models, dens = case["base"], case["models"], case["density"]
        tv = torch.stack([m - base for m in models]).to(DEV).contiguous()
        got = ops.pcb_vectors(tv, dens).cpu()
        ref = case["pcb"]
        assert torch.allclose(got, ref, rtol=2e-4, atol=1e-9), ((got - ref).abs().max(), ref.abs().max())


def test_load_merging_module_pcb():
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module

    g2 = load_golden("g2_merger.pt")
    mm = load_merging_module(MergeType.PCB, LearnType.TASK_WISE, _tiny_model(g2["cfg"]), g2["pretrain"], [dict(f) for f in g2["finetunes"]],
                             set(), ties_density=0.2, disable_softmax=True)
    pre, al = O.align_state_dicts(g2["pretrain"], g2["finetunes"])
    base, _ = O.flatten_model(pre)
    want = O.pcb_vectors(base, [O.flatten_model(f)[0] for f in al], 0.2)
    got = mm.compact_task_vectors().cpu()
    assert torch.allclose(got, want, rtol=2e-4, atol=1e-9), (got - want).abs().max()


@pytest.mark.parametrize("kind", ["blair_large", "recformer_large"])
def test_large_configs_match_oracle(kind):
    """24 x 1024, 16 heads (BLAIR_LARGE / RECFORMER_LARGE, BASELINE config 5's encoder): every GEMM mode against the CPU oracle"""
    import sys

    sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent / "tools"))
    import large_models_check as L
    from mergerec_amd.engine import EncoderSpec

    res = L.check(kind, getattr(EncoderSpec, kind)(), verbose=False)
    assert res["f32"][0] <= 5e-6 and res["bf16x6"][0] <= 5e-6 and res["f16x3"][0] <= 5e-6 and res["bf16x3"][0] <= 5e-5, res
    assert all(v[1] <= 1e-4 for v in res.values()), res  # the path's contract: cosine logits within 1e-4


# ------------------------------------------------------------------ boundary: test_model called exactly as merge_test.py:91-110 calls it
def test_test_model_called_like_the_reference(tmp_path):
    """The reference's call site, argument for argument (keywords), on a dataset directory in its JSON format with a local tokenizer;
    outputs in the reference's shapes: metric_dict keys test/dataset_{i}/..., CSV indexed by the directory name."""
    import csv

    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.utils import test_model
    from tests.conftest import GOLDEN

    config = SimpleNamespace(
        model_type="BLAIR_BASE", data_paths=[GOLDEN / "mini_dataset"], batch_size=8, max_seq_len=96, max_attribute_len=12, max_items=20,
        num_workers=0, sequence_prompt=None, item_prompt=None, reverse_sequence=True, precision="32-true", test_data_split="test",
        metrics_path=tmp_path / "m.csv", predictions_path=tmp_path / "p.pt", item_embeddings_path=tmp_path / "i.pt",
        user_embeddings_path=tmp_path / "u.pt", metric_names=["NDCG", "RECALL"], ks=[1, 5, 10, 50], similarity="cosine")
    model = ModelType[config.model_type].value(
        model_name_or_path=None, tokenizer_name_or_path=str(GOLDEN / "mini_tokenizer"), lora_config=None, pooling_method="cls",
        model_kwargs={"init_seed": 7, "spec_overrides": dict(hidden=128, heads=2, layers=2, intermediate=256), "device": DEV}, tokenizer_kwargs={})
    module = RecModule(model=model, evaluator=Evaluator(metrics=config.metric_names, ks=config.ks), negative_sample=None, similarity=config.similarity)
    _, metrics, scores, labels = test_model(
        module=module,
        model_type=ModelType[config.model_type],
        data_paths=config.data_paths,
        model_tokenizer=model.tokenizer,
        batch_size=config.batch_size,
        max_seq_len=config.max_seq_len,
        max_attribute_len=config.max_attribute_len,
        max_items=config.max_items,
        num_workers=config.num_workers,
        sequence_prompt=config.sequence_prompt,
        item_prompt=config.item_prompt,
        reverse_sequence=config.reverse_sequence,
        precision=config.precision,
        data_split=config.test_data_split,
        metrics_path=config.metrics_path,
        predictions_path=config.predictions_path,
        item_embeddings_path=config.item_embeddings_path,
        user_embeddings_path=config.user_embeddings_path,
    )
    assert len(metrics) == 1 and set(metrics[0]) >= {"test/NDCG@10", "test/Recall@50", "test/loss"}
    n_users, n_items = labels[0].numel(), torch.load(config.item_embeddings_path)[0].shape[0]
    assert scores[0].shape == (n_users, n_items)  # predictions_path given -> the (users, items) block is materialised like the reference's
    rows = list(csv.DictReader(open(config.metrics_path)))
    assert rows[0]["dataset"] == "mini_dataset" and float(rows[0]["test/NDCG@10"]) == metrics[0]["test/NDCG@10"]
    pred = torch.load(config.predictions_path)["mini_dataset"]
    assert torch.equal(pred["scores"], scores[0]) and torch.equal(pred["labels"], labels[0])
    # the reference's top-k of those scores ranks the labels where the fused kernel did
    top = torch.topk(scores[0], min(50, n_items), dim=1).indices
    hit10 = (top[:, :10] == labels[0][:, None]).any(1).double().mean().item()
    assert abs(hit10 - metrics[0]["test/Recall@10"]) < 1e-12
    with pytest.raises(ValueError):
        test_model(module, ModelType.BLAIR_BASE, config.data_paths, model.tokenizer, 8, 96, 12, 20, 0, None, None, True, "32-true", "train")


def test_input_contract_violations_surface_at_check_inputs():
    """ids / token types / item positions out of range, an unattended CLS and foreign global-attention patterns are caught inside the
    packing kernel (no host sync per batch) and raised by check_inputs(); a bad id can never fault the gather (indices are clamped)."""
    from mergerec_amd.engine import InputError
    from mergerec_amd.module import ModelType

    over = dict(hidden=128, heads=2, layers=1, intermediate=128, vocab=100, max_pos=64)
    blair = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 1, "spec_overrides": over, "device": DEV})
    ids = torch.randint(3, 100, (4, 20))
    mask = torch.ones(4, 20, dtype=torch.int64)
    mask[2, 7:] = 0
    ok = {"input_ids": ids.to(DEV), "attention_mask": mask.to(DEV)}
    blair.encode_normalized(ok, normalize=True)
    blair.check_inputs()  # clean
    bad = dict(ok, input_ids=ids.clone().index_put_((torch.tensor([1]), torch.tensor([3])), torch.tensor(100)).to(DEV))
    out = blair.encode_normalized(bad, normalize=True)  # runs (clamped gather), flagged
    assert torch.isfinite(out).all()
    with pytest.raises(InputError, match="input_ids"):
        blair.check_inputs()
    blair.check_inputs()  # the flag was cleared
    nocls = dict(ok, attention_mask=mask.clone().index_put_((torch.tensor([0]), torch.tensor([0])), torch.tensor(0)).to(DEV))
    blair.encode_normalized(nocls, normalize=True)
    with pytest.raises(InputError, match="CLS"):
        blair.check_inputs()
    with pytest.raises(InputError):  # "now": checked at once
        blair.runner.pack(bad, DEV, validate="now")
    blair.runner.pack(bad, DEV, validate=False)
    blair.check_inputs()  # unchecked packs leave no trace
    with pytest.raises(ValueError, match="position table"):
        blair.encode_normalized({"input_ids": torch.zeros(1, 80, dtype=torch.int64, device=DEV), "attention_mask": torch.ones(1, 80, dtype=torch.int64, device=DEV)}, True)
    # stale host lengths (rows edited after .to()) are a length mismatch, not silent garbage
    from mergerec_amd.model_batch import BatchItem

    moved = BatchItem(items={"input_ids": ids, "attention_mask": mask}).to(DEV).items
    moved["attention_mask"][3, 10:] = 0
    blair.encode_normalized(moved, normalize=True)
    with pytest.raises(InputError, match="lengths"):
        blair.check_inputs()

    rec = ModelType.RECFORMER_BASE.value(model_kwargs={"init_seed": 1, "spec_overrides": dict(over, max_pos=128), "device": DEV})
    tt = torch.randint(0, 4, (4, 20))
    ip = torch.randint(0, 51, (4, 20))
    gm = torch.zeros(4, 20, dtype=torch.int64)
    gm[:, 0] = 1
    rb = {"input_ids": ids.to(DEV), "attention_mask": mask.to(DEV), "token_type_ids": tt.to(DEV), "item_position_ids": ip.to(DEV),
          "global_attention_mask": gm.to(DEV)}
    rec.encode_normalized(rb, normalize=True)
    rec.check_inputs()
    for key, value, msg in (("token_type_ids", 4, "token_type"), ("item_position_ids", 51, "item_position"), ("global_attention_mask", 1, "global attention")):
        t = {"token_type_ids": tt, "item_position_ids": ip, "global_attention_mask": gm}[key].clone()
        t[1, 2] = value
        rec.encode_normalized(dict(rb, **{key: t.to(DEV)}), normalize=True)
        with pytest.raises(InputError, match=msg):
            rec.check_inputs()


def test_ties_vectors_at_full_model_size_match_the_reference():
    """8(f).1 at size: the reference's get_ties_vectors on BLaIR-base's flat length (P = 124,645,632, 8 models, density 0.2; fixture g17 from
    oracle/gen_golden_ties_fullsize.py) against the device pipeline -- radix select of the 24.9 M largest magnitudes per model, sign
    election, disjoint mean.  The reference trims with torch.topk(|tau|, k).indices (algorithms/ties.py:21), whose choice among magnitudes
    that tie EXACTLY at the k-th value is unspecified; the device rule is "lowest indices".  At this size such ties exist, so the
    comparison is exact everywhere except at positions where some model's |tau| equals its own threshold: sampled values bit for bit
    elsewhere, survivor counts and float64 sums within what those positions can change."""
    from mergerec_amd import ops

    fx = load_golden("g17_ties_fullsize.pt")
    P, N = fx["P"], fx["N"]
    g = torch.Generator().manual_seed(fx["seed"])
    base = torch.randn(P, generator=g) * 0.02
    tv = torch.empty(N, P, dtype=torch.float32, device=DEV)
    for i in range(N):  # (one generator: base first, then the models in order)
        tv[i] = ((base + torch.randn(P, generator=g) * 1e-3) - base).to(DEV)
    k = int(fx["density"] * P)
    ambiguous = torch.zeros(P, dtype=torch.bool, device=DEV)   # positions where torch.topk's tie choice can differ from the device rule
    thr_max, n_amb = 0.0, 0
    thr = torch.empty(1, dtype=torch.float32, device=DEV)
    for i in range(N):
        a = tv[i].abs()
        ops.kth_largest_value(tv[i], k, False, thr)
        at_thr = a == thr
        if int(at_thr.sum()) != k - int((a > thr).sum()):        # more elements tie at the threshold than the cut admits
            ambiguous |= at_thr
            n_amb += int(at_thr.sum())
            thr_max = max(thr_max, float(thr))
        ops.abs_topk_mask(tv[i], k, out=tv[i])
        assert int((tv[i] != 0).sum()) <= k
    got = ops.ties_combine(tv)
    pos = fx["sample_pos"].to(DEV)
    clear = ~ambiguous[pos]
    for i in range(N):
        nnz = int((got[i] != 0).sum())
        assert abs(nnz - fx["nnz"][i]) <= n_amb, (i, nnz, fx["nnz"][i], n_amb)
        assert torch.equal(got[i][pos][clear].cpu(), fx["sample"][i][clear.cpu()]), i
        s, ab = float(got[i].double().sum()), float(got[i].double().abs().sum())
        slack = n_amb * thr_max + 1e-9 * fx["abs_sum"][i]
        assert abs(s - fx["sum"][i]) <= slack and abs(ab - fx["abs_sum"][i]) <= slack, (i, s, fx["sum"][i], slack)
    assert n_amb < 1000, n_amb  # a handful of positions out of 8 x 125 M
    print(f"[ties full size] positions tied at a trimming threshold: {n_amb}; survivors per model equal to the reference's within that; "
          f"{int(clear.sum())} of {pos.numel()} sampled positions compared bit for bit")


def test_lns_and_pcb_vectors_at_full_model_size_match_the_reference():
    """8(f).1 at size (fixture g18, oracle/gen_golden_lns_pcb_fullsize.py; same inputs as g17): Localize-and-Stitch (8 models, density 0.05)
    and PCB (4 models, density 0.2) on P = 124,645,632.  L&S trims with torch.topk like TIES, so positions on a model's trimming threshold
    are excluded from the exact comparison; PCB is a chain of sorts, clamps, exp and tanh in fp32 -- compared to the tolerance of the g6
    test, relative to the largest entry."""
    from mergerec_amd import ops

    fx = load_golden("g18_lns_pcb_fullsize.pt")
    P, N = fx["P"], fx["N"]
    g = torch.Generator().manual_seed(fx["seed"])
    base = torch.randn(P, generator=g) * 0.02
    tv = torch.empty(N, P, dtype=torch.float32, device=DEV)
    for i in range(N):
        tv[i] = ((base + torch.randn(P, generator=g) * 1e-3) - base).to(DEV)
    pos = fx["sample_pos"].to(DEV)
    # ---- PCB first (it reads the untouched task vectors of the first models)
    n_pcb = fx["pcb_models"]
    got = ops.pcb_vectors(tv[:n_pcb].contiguous(), fx["pcb_density"])
    ref = fx["pcb"]
    scale = float(ref["sample"].abs().max())
    for i in range(n_pcb):
        assert float((got[i][pos].cpu() - ref["sample"][i]).abs().max()) <= 2e-4 * scale, i
        assert abs(float(got[i].double().abs().sum()) - ref["abs_sum"][i]) <= 2e-4 * ref["abs_sum"][i], i
        assert abs(int((got[i] != 0).sum()) - ref["nnz"][i]) <= 1e-4 * P, (i, int((got[i] != 0).sum()), ref["nnz"][i])
    del got
    # ---- Localize-and-Stitch
    k = int(fx["lns_density"] * P)
    masks = torch.empty(N, P, dtype=torch.uint8, device=DEV)
    ambiguous = torch.zeros(P, dtype=torch.bool, device=DEV)
    thr = torch.empty(1, dtype=torch.float32, device=DEV)
    n_amb, thr_max = 0, 0.0
    scratch = torch.empty(P, dtype=torch.float32, device=DEV)
    for i in range(N):
        a = tv[i].abs()
        ops.kth_largest_value(tv[i], k, False, thr)
        at_thr = a == thr
        if int(at_thr.sum()) != k - int((a > thr).sum()):
            ambiguous |= at_thr
            n_amb += int(at_thr.sum())
            thr_max = max(thr_max, float(thr))
        _, m = ops.abs_topk_mask(tv[i], k, out=scratch, want_mask=True)
        masks[i] = m
    got = ops.lns_combine(tv, masks)
    ref = fx["lns"]
    clear = ~ambiguous[pos]
    for i in range(N):
        assert abs(int((got[i] != 0).sum()) - ref["nnz"][i]) <= n_amb, (i, n_amb)
        assert torch.equal(got[i][pos][clear].cpu(), ref["sample"][i][clear.cpu()]), i
        slack = n_amb * thr_max + 1e-9 * ref["abs_sum"][i]
        assert abs(float(got[i].double().sum()) - ref["sum"][i]) <= slack and abs(float(got[i].double().abs().sum()) - ref["abs_sum"][i]) <= slack, i
    assert n_amb < 1000, n_amb
    print(f"[lns / pcb full size] L&S: {n_amb} positions tied at a trimming threshold, {int(clear.sum())} of {pos.numel()} sampled positions bit for bit; "
          f"PCB: sampled values within 2e-4 of the largest entry")
