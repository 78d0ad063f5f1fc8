"""Pins oracle/ref_cpu.py against outputs of the reference itself (tests/golden, made by
oracle/gen_golden.py in the build container).  CPU only."""
import math
from collections import OrderedDict

import pytest
import torch

from oracle import ref_cpu as O
from tests.conftest import load_golden


def _cfg(d):
    return O.EncoderConfig(**d)


# ------------------------------------------------------------------ G1: evaluator (a18, a19)
def test_evaluator_matches_reference():
    for case in load_golden("g1_evaluator.pt"):
        got = O.evaluate(case["scores"], case["labels"], ["NDCG", "RECALL"], case["ks"], "test/")
        assert list(got.keys()) == case["metric_key_order"]
        for k, v in case["metrics"].items():
            assert got[k] == v, (k, got[k], v)  # same float32 gain table, same Python float sum
        # canonical top-k == torch.topk as a multiset of (score) and exactly where no ties exist
        val, idx = O.topk_canonical(case["scores"], case["ref_topk_idx"].shape[1])
        assert torch.equal(val, case["ref_topk_val"])
        s = case["scores"]
        for r in range(s.shape[0]):
            v = val[r]
            if len(torch.unique(v)) == len(v):
                assert torch.equal(idx[r], case["ref_topk_idx"][r])
            # canonical: ties broken by ascending index
            same = v[1:] == v[:-1]
            assert bool((idx[r][1:][same] > idx[r][:-1][same]).all())


def test_topk_canonical_nan_first():
    s = torch.tensor([[0.1, float("nan"), 0.5, 0.5, float("nan"), -1.0]])
    val, idx = O.topk_canonical(s, 4)
    assert idx.tolist() == [[1, 4, 2, 3]]


# ------------------------------------------------------------------ G2: merger (a2..a7)
def _alpha(case, key, n):
    w = case["weights"]
    gw = torch.tensor(w["global_weights"][key])
    gb = torch.tensor(w["global_biases"][key])
    per = torch.tensor(w["per_weights"][key])[:n]  # _base.py:72 truncation
    return O.effective_alpha(gw, gb, per, disable_softmax=not case["use_softmax"])


def test_merge_matches_reference_bitwise():
    g2 = load_golden("g2_merger.pt")
    pre, fts = O.align_state_dicts(g2["pretrain"], g2["finetunes"])
    assert "item_embeddings" not in pre
    base, shape_dict = O.flatten_model(pre)
    models = [O.flatten_model(ft)[0] for ft in fts]
    tv = O.get_task_vectors(base, models)
    assert float(base.double().sum()) == g2["base_flat_checksum"]
    assert torch.equal(tv, g2["tv_flat"])
    n = tv.shape[0]
    for case in g2["cases"]:
        assert list(shape_dict.keys()) == case["shape_keys"]
        assert [tuple(s) for s in shape_dict.values()] == case["shapes"]
        if case["learn_type"] == "TASK_WISE":
            assert case["groups"] == ["all"]
            merged = O.merge_task_wise(base, tv, _alpha(case, "all", n))
            init = O.merge_task_wise(base, tv, O.effective_alpha(torch.tensor([1.0]), torch.tensor([0.0]), torch.full((n,), 0.3), not case["use_softmax"]))
        else:
            groups = O.group_parameters_by_layer(shape_dict)
            assert list(groups.keys()) == case["groups"]
            merged = O.merge_layer_wise(base, tv, groups, {k: _alpha(case, k, n) for k in groups})
            a0 = O.effective_alpha(torch.tensor([1.0]), torch.tensor([0.0]), torch.full((n,), 0.3), not case["use_softmax"])
            init = O.merge_layer_wise(base, tv, groups, {k: a0 for k in groups})
        assert torch.equal(merged, case["merged_flat"]), case["learn_type"]
        if case["init_merged_flat"] is not None:
            assert torch.equal(init, case["init_merged_flat"])
        # a6: named views
        sd = O.get_state_dict(merged, shape_dict)
        assert list(sd.keys()) == case["shape_keys"]
        # G5: the reference module's own forward (re-merge + functional HF model) -> CLS
        cfg = _cfg(g2["cfg"])
        cls = O.roberta_encode(sd, g2["input_ids"], g2["attention_mask"], cfg, prefix="model.")
        assert torch.allclose(cls, case["cls"], atol=2e-5, rtol=1e-5), (cls - case["cls"]).abs().max()


def test_model_merger_fixed_weight_merges_match_reference_bitwise():
    """ModelMerger.merge("task_vector" | "linear") of the reference (merger.py:46-93): running sums in model order"""
    g2 = load_golden("g2_merger.pt")
    pre = g2["pretrain"]
    base, _ = O.flatten_model(pre)
    models = [O.flatten_model(OrderedDict((k, ft[k]) for k in pre))[0] for ft in g2["finetunes"]]
    assert torch.equal(O.merge_running(base, models, [0.5, 0.25, 0.7]), g2["model_merger_task_vector"])
    assert torch.equal(O.merge_running(None, models, [0.2, 0.3, 0.5]), g2["model_merger_linear"])
    # the learnable-alpha module sums the products first and adds the base last: same value, different rounding
    tw = O.merge_task_wise(base, O.get_task_vectors(base, models), torch.tensor([0.5, 0.25, 0.7]))
    assert not torch.equal(tw, g2["model_merger_task_vector"]) and torch.allclose(tw, g2["model_merger_task_vector"], atol=1e-6)


def test_flatten_promotes_int_buffer():
    g2 = load_golden("g2_merger.pt")
    sd = OrderedDict([("model.embeddings.position_ids", torch.arange(10).view(1, 10))] + list(g2["pretrain"].items()))
    flat, _ = O.flatten_model(sd)
    assert str(flat.dtype) == g2["flatten_with_int_buffer_dtype"] == "torch.float32"
    assert torch.equal(flat[:12], g2["flatten_with_int_buffer_head"])


# ------------------------------------------------------------------ G3: RoBERTa (a8, a11, a13)
def test_roberta_matches_library():
    g3 = load_golden("g3_roberta.pt")
    cfg = _cfg(g3["cfg"])
    cls, hidden = O.roberta_encode(g3["state_dict"], g3["input_ids"], g3["attention_mask"], cfg, "model.", return_hidden=True)
    m = g3["attention_mask"].bool()
    for h, ref in zip(hidden, g3["hidden_states"]):
        assert torch.allclose(h[m], ref[m], atol=3e-5, rtol=1e-5), (h[m] - ref[m]).abs().max()
    assert torch.allclose(cls, g3["cls"], atol=3e-5, rtol=1e-5)


def test_roberta_true_dims_single_layer():
    big = load_golden("g3_roberta.pt")["big"]
    cfg = _cfg(big["cfg"])
    sd = O.random_state_dict(O.roberta_param_shapes(cfg), seed=big["seed"], std=big["std"])
    assert abs(float(sum(v.double().sum() for v in sd.values())) - big["checksum"]) < 1e-6, "seeded weight generator drifted"
    _, hidden = O.roberta_encode(sd, big["input_ids"], big["attention_mask"], cfg, "model.", return_hidden=True)
    m = big["attention_mask"].bool()
    assert torch.allclose(hidden[0][m], big["emb"][m], atol=1e-5)
    assert torch.allclose(hidden[-1][m], big["last"][m], atol=2e-5), (hidden[-1][m] - big["last"][m]).abs().max()


# ------------------------------------------------------------------ G4: Recformer (a9, a10, a12)
def test_recformer_matches_reference():
    g4 = load_golden("g4_recformer.pt")
    for case in g4["cases"]:
        cfg = _cfg(case["cfg"])
        b = case["batch"]
        cls, hidden = O.recformer_encode(
            case["state_dict"], b["input_ids"], b["attention_mask"], b["global_attention_mask"], b["token_type_ids"],
            b["item_position_ids"], cfg, "model.", return_hidden=True,
        )
        m = b["attention_mask"].bool()
        assert torch.allclose(hidden[0][m], case["emb"][m], atol=1e-5)
        for li, (h, ref) in enumerate(zip(hidden, case["hidden_states"])):
            assert torch.allclose(h[m], ref[m], atol=5e-5, rtol=1e-5), (li, (h[m] - ref[m]).abs().max())
        assert torch.allclose(cls, case["cls"], atol=5e-5, rtol=1e-5)


def test_position_ids():
    ids = torch.tensor([[0, 5, 1, 7, 1, 1], [0, 2, 1, 1, 1, 1]])
    assert O.position_ids_from_input_ids(ids, 1).tolist() == [[2, 3, 1, 4, 1, 1], [2, 3, 1, 1, 1, 1]]


# ------------------------------------------------------------------ G6: TIES / Localize-and-Stitch / PCB (8(f).1)
def test_taskvector_algorithms_match_reference():
    for case in load_golden("g6_taskvector_algos.pt")["cases"]:
        base, models, dens = case["base"], case["models"], case["density"]
        assert torch.equal(O.ties_vectors(base, models, dens), case["ties"])
        assert torch.equal(O.localize_and_stitch_vectors(base, models, dens), case["lns"])
        got = O.pcb_vectors(base, models, dens)
        assert torch.allclose(got, case["pcb"], rtol=1e-5, atol=1e-9), (got - case["pcb"]).abs().max()


def test_distill_losses_match_reference():
    """oracle restatement of loss_fn.py vs the values and autograd gradients the reference's own classes produced"""
    g7 = load_golden("g7_distill_losses.pt")
    seen = set()
    for c in g7["cases"]:
        z = c["z"].clone().requires_grad_(True)
        loss = O.distill_loss(c["loss"], z, c["t"], c["temperature"], c["coefficient"], c["margin"])
        (grad,) = torch.autograd.grad(loss, z)
        assert torch.allclose(loss, c["value"], rtol=1e-6, atol=1e-7), (c["loss"], loss, c["value"])
        assert torch.allclose(grad, c["grad"], rtol=1e-5, atol=1e-9), (c["loss"], (grad - c["grad"]).abs().max())
        seen.add(c["loss"])
    assert seen == set(O.DISTILL_LOSSES)


def test_forward_distill_matches_reference():
    f = load_golden("g7_distill_losses.pt")["forward_distill"]
    reps = f["reps"].clone().requires_grad_(True)
    loss = O.forward_distill(reps, f["item_embeddings"], f["score_embeddings"], f["dataset_indexes"], f["sequence_ids"],
                             lambda z, t: O.distill_loss(f["loss"], z, t, f["temperature"], f["coefficient"]))
    (g,) = torch.autograd.grad(loss, reps)
    assert torch.allclose(loss, f["value"], rtol=1e-6)
    assert torch.allclose(g, f["rep_grad"], rtol=1e-5, atol=1e-8)


# ---------------------------------------------------------------------------------------------- fine-tuning (g9: the reference's RecModule)
def test_negative_sample_scores_match_reference():
    """module.py:79-131, 183: scores / labels / loss of the three negative-sampling modes"""
    for c in load_golden("g9_finetune.pt")["scores"]:
        scores, labels = O.negative_sample_scores(c["user"], c["target"], c["negatives"], c["mode"], c["k"])
        assert torch.equal(labels, c["labels"]), c["mode"]
        torch.testing.assert_close(scores, c["scores"], rtol=0, atol=2e-7)
        torch.testing.assert_close(O.finetune_loss(scores, labels, c["temperature"]), c["loss"], rtol=1e-6, atol=1e-6)


def test_optimizer_groups_schedule_and_adamw_match_reference():
    """module.py:44-72 driven like Lightning drives it: groups, warm-up schedule, clip, AdamW -- replayed from the recorded gradients"""
    for c in load_golden("g9_finetune.pt")["optim"]:
        names = list(c["init"].keys())
        wd = O.optimizer_groups(names, c["weight_decay"])
        for grp_names, grp_wd in zip(c["group_names"], c["group_weight_decay"]):
            assert all(wd[n] == grp_wd for n in grp_names), (grp_names, grp_wd)
        warm = O.resolve_warmup(c["warmup_steps"], c["estimated_stepping_batches"])
        p = OrderedDict((n, t.clone()) for n, t in c["init"].items())
        m = OrderedDict((n, torch.zeros_like(t)) for n, t in p.items())
        v = OrderedDict((n, torch.zeros_like(t)) for n, t in p.items())
        for s, rec in enumerate(c["steps"]):
            lr = c["learning_rate"] * O.linear_warmup_multiplier(s, warm, c["estimated_stepping_batches"])
            assert all(abs(lr - x) <= 1e-12 for x in rec["lr"]), (s, lr, rec["lr"])
            coef = 1.0
            if c["gradient_clip_val"] is not None:
                coef = O.clip_coefficient(list(rec["grads"].values()), c["gradient_clip_val"])
            for n in names:
                O.adamw_step(p[n], rec["grads"][n] * coef, m[n], v[n], lr, wd[n], s + 1, c["betas"], c["eps"])
                torch.testing.assert_close(p[n], rec["params"][n], rtol=1e-6, atol=1e-7, msg=lambda e: f"step {s} {n}: {e}")


def test_whole_training_step_matches_reference_on_hf_roberta():
    """g9 roberta_step: the reference's RecModule.training_step around transformers' RobertaModel (in-batch negatives, cosine, T = 0.05):
    loss and d loss / d every parameter, against torch autograd through the oracle's encoder restatement"""
    st = load_golden("g9_finetune.pt")["roberta_step"]
    cfg = O.EncoderConfig(**{k: v for k, v in st["cfg"].items() if k in O.EncoderConfig.__dataclass_fields__})
    p = OrderedDict((k, v.clone().requires_grad_(v.is_floating_point())) for k, v in st["state_dict"].items())
    enc = lambda b: O.maybe_normalize(O.roberta_encode(p, b["input_ids"], b["attention_mask"], cfg, prefix="model."))
    scores, labels = O.negative_sample_scores(enc(st["sequence"]), enc(st["target"]), None, "IN_BATCH", None)
    loss = O.finetune_loss(scores, labels, st["temperature"])
    torch.testing.assert_close(loss.detach(), st["loss"], rtol=1e-5, atol=1e-5)
    loss.backward()
    gmax = max(float(g.abs().max()) for g in st["grads"].values() if g is not None)
    for k, g in st["grads"].items():
        if g is None:  # pooler: not on the CLS path
            assert p[k].grad is None or float(p[k].grad.abs().max()) == 0.0, k
            continue
        err = float((p[k].grad - g).abs().max())
        assert err <= 2e-4 * max(float(g.abs().max()), 1e-3 * gmax), (k, err)


def _g10_finetunes(g10):
    fts = [O.perturbed_state_dict(g10["pretrain"], seed=s, std=g10["finetune_std"]) for s in g10["finetune_seeds"]]
    assert abs(float(sum(v.double().sum() for ft in fts for v in ft.values())) - g10["finetune_checksum"]) < 1e-6 * abs(g10["finetune_checksum"]) + 1e-6
    return fts


@pytest.mark.parametrize("case", [0, 1])
def test_whole_merge_train_step_matches_reference(case):
    """g10: one collaborative-merging step of the reference itself (load_merging_module + DistillSequenceModule.training_step around
    transformers' RobertaModel, SINGLE_PSEUDO_LABEL_KD) -- loss and d loss / d alpha -- against autograd through the oracle's merge +
    encoder + loss restatements"""
    g10 = load_golden("g10_merge_train_step.pt")
    c = g10["cases"][case]
    cfg = O.EncoderConfig(**{k: v for k, v in g10["cfg"].items() if k in O.EncoderConfig.__dataclass_fields__})
    pre, fts = g10["pretrain"], _g10_finetunes(g10)
    pre_a, fts_a = O.align_state_dicts(pre, fts)
    base, shapes = O.flatten_model(pre_a)
    tv = O.get_task_vectors(base, [O.flatten_model(ft)[0] for ft in fts_a])
    N = len(fts)
    groups = O.group_parameters_by_layer(shapes) if c["learn_type"] == "LAYER_WISE" else None
    keys = c["groups"]
    per = {k: torch.full((N,), g10["initial_per_weight"], requires_grad=True) for k in keys}
    gw = {k: torch.ones(1, requires_grad=True) for k in keys}
    gb = {k: torch.zeros(1, requires_grad=True) for k in keys}
    if groups is None:
        merged = O.merge_task_wise(base, tv, O.effective_alpha(gw["all"], gb["all"], per["all"], True))
    else:
        assert list(groups.keys()) == keys
        merged = O.merge_layer_wise(base, tv, groups, {k: O.effective_alpha(gw[k], gb[k], per[k], True) for k in keys})
    sd = O.get_state_dict(merged, shapes)
    reps = O.maybe_normalize(O.roberta_encode(sd, g10["input_ids"], g10["attention_mask"], cfg, prefix="model."))
    loss = O.forward_distill(reps, g10["item_embeddings"], g10["score_embeddings"], g10["dataset_indexes"], g10["sequence_ids"],
                             lambda z, t: O.distill_loss("SINGLE_PSEUDO_LABEL_KD", z, t, g10["temperature"], g10["coefficient"]))
    torch.testing.assert_close(loss.detach(), c["loss"], rtol=2e-5, atol=2e-5)
    loss.backward()
    for name, mine in (("per_weights", per), ("global_weights", gw), ("global_biases", gb)):
        for k in keys:
            want = c["grads"][name][k]
            scale = max(float(want.abs().max()), 1e-3)
            assert float((mine[k].grad - want).abs().max()) <= 2e-3 * scale, (name, k, mine[k].grad, want)


def test_recformer_parameter_gradients_match_reference():
    """g11: d sum(normalize(CLS) * R) / d every parameter of the reference's RecformerModel (its embeddings / mask helpers driving the
    library's LongformerEncoder, as for g4) against autograd through the oracle's restatement"""
    g4, g11 = load_golden("g4_recformer.pt"), load_golden("g11_recformer_grads.pt")
    for case, gr in zip(g4["cases"], g11["cases"]):
        cfgd, sd, b = case["cfg"], case["state_dict"], case["batch"]
        cfg = O.EncoderConfig(**{k: cfgd[k] for k in cfgd if k in O.EncoderConfig.__dataclass_fields__})
        p = OrderedDict((k, v.clone().float().requires_grad_(v.is_floating_point())) for k, v in sd.items())
        cls = O.recformer_encode(p, b["input_ids"], b["attention_mask"], b["global_attention_mask"], b["token_type_ids"], b["item_position_ids"], cfg,
                                 prefix="model.")
        (O.maybe_normalize(cls) * gr["R"]).sum().backward()
        gmax = max(float(g.abs().max()) for g in gr["grads"].values() if g is not None)
        for k, g in gr["grads"].items():
            if g is None:
                assert p[k].grad is None or float(p[k].grad.abs().max()) == 0.0, k
                continue
            err = float((p[k].grad - g).abs().max())
            assert err <= 5e-4 * max(float(g.abs().max()), 1e-3 * gmax), (k, err, float(g.abs().max()))


def test_oracle_mean_pooling_matches_transformers_padded_mean():
    """encoder/_base.py:42-43: pooling "mean" = last_hidden_state.mean(dim=1) over the padded length; g3 stores the padded hidden states
    transformers produced (pad positions included).  Also the fact the HIP path builds on: every pad position of a sequence holds the same
    vector (one extra query row per sequence reproduces them)."""
    import torch
    from oracle import ref_cpu as O
    from tests.conftest import load_golden

    g3 = load_golden("g3_roberta.pt")
    cfgd = g3["cfg"]
    cfg = O.EncoderConfig(**{k: cfgd[k] for k in cfgd if k in O.EncoderConfig.__dataclass_fields__})
    got = O.roberta_encode(g3["state_dict"], g3["input_ids"], g3["attention_mask"], cfg, prefix="model.", pooling="mean")
    want = g3["hidden_states"][-1].mean(dim=1)
    assert float((got - want).abs().max()) <= 1e-5
    h, lens = g3["hidden_states"][-1], g3["attention_mask"].sum(1)
    for b in range(h.shape[0]):
        n = int(lens[b])
        if n < h.shape[1]:
            assert float((h[b, n:] - h[b, n]).abs().max()) == 0.0
