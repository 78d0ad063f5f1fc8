"""north_star: "NDCG@10 within 1e-3 of reference across all 8 Amazon domains" -- BASELINE configs[3]'s model (ONE 8-domain task-vector merge
of BLaIR-base at true dimensions) evaluated on every domain's full catalog (4,968 ... 27,932 items, 114,075 in all), 4,096 users per domain.

Fixture: tests/golden/g13_8domain_blair_base.pt, produced in the build container by oracle/gen_golden_8domain.py from the reference itself
(its load_merging_module / get_state_dict, transformers' RobertaModel, user @ item.T, its Evaluator; CPU, fp32).  Inputs are regenerated from
seeds here.  Checked per domain, in the arithmetic the reference's default precision flag selects (bf16-mixed -> f16x3) through the drop-in evaluation loop: user embeddings, sampled
item rows and their logits within 3e-6 (north_star: 1e-4); the ranked top-50 equal to the reference's wherever its scores are more than twice the measured
logit error apart; label ranks equal up to
near-ties; every Recall / NDCG value within 1e-3; the loss within 1e-3."""

import os

import pytest
import torch

from oracle import ref_cpu as O
from tests.conftest import load_golden, prefetched, register_prefetch, seeded_state_dicts

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOGIT_TOL = 1e-4       # north_star
LOGIT_BOUND = 3e-6     # what fp32 summation order costs on these init-like weights (r02-r04 measured 1.6-2.3e-6 in every arithmetic): asserted on
                       # every logit the fixture lets us compare -- a sampled block, every user's reference top-52, every label
# Near-tie budget: DERIVED per domain from the logit error measured in that run (r04; ADVICE r03 -- through r03 it was a constant moved from
# 2e-6 to 2.5e-6 after an observed crossing): if every logit is within eps of the reference's, two items can swap places only when their
# REFERENCE scores are within 2 eps of each other.  Every differing rank position / label move is verified to be such a pair.
NDCG_TOL = 1e-3        # north_star


def _build_state_dicts():
    """host-only: the fixture and its pretrained + 8 fine-tuned BLaIR-base state dicts (the order the reference's wrapper yielded:
    perturbations are drawn along it)"""
    fx = load_golden("g13_8domain_blair_base.pt")
    cfg = O.EncoderConfig()
    pre, fts = seeded_state_dicts(O.roberta_param_shapes(cfg), fx["key_order"], fx["seed_pre"], 0.02, fx["pre_checksum"], fx["seed_ft"], fx["ft_std"])
    return fx, cfg, pre, fts


register_prefetch("g13", _build_state_dicts, match=("test_8domain_gpu.py",))


@pytest.fixture(scope="module")
def merged():
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.module import ModelType

    fx, cfg, pre, fts = prefetched("g13")  # host-only part, drawn in the background (tests/_prefetch.py)
    model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 0, "device": DEV})
    model.load_state_dict(pre)
    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, model, pre, fts, set(), disable_softmax=True)  # merge_test.py:35-71
    mm.load_weights_from_dict({"global_weights": {"all": [1.0]}, "global_biases": {"all": [0.0]}, "per_weights": {"all": list(fx["alphas"])}})
    sd = {k: v.detach().clone() for k, v in mm.get_state_dict().items()}
    merged_sum = float(sum(v.double().sum() for v in sd.values()))
    assert abs(merged_sum - fx["merged_checksum"]) < 1e-9 * max(1.0, abs(fx["merged_checksum"])) + 1e-5, (merged_sum, fx["merged_checksum"])
    del mm, model, fts
    torch.cuda.empty_cache()
    model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 0, "device": DEV})  # product default arithmetic
    model.load_state_dict(sd)
    return fx, cfg, model


def test_all_eight_domains_match_the_reference(merged, tmp_path):
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.model_batch import BatchSequence
    from mergerec_amd.module import RecModule
    from mergerec_amd.synthetic import CATALOG_SIZES, make_domain
    from mergerec_amd.utils import test_model_on_dataloaders

    fx, cfg, model = merged
    assert list(fx["domains"]) == list(CATALOG_SIZES), "the fixture covers every domain"
    module = RecModule(model=model, evaluator=Evaluator(["NDCG", "RECALL"], fx["ks"]), similarity="cosine")
    worst = dict(logit=0.0, ndcg10=0.0, any_metric=0.0)
    only = os.environ.get("MR_8DOM_ONLY")  # debugging aid: a comma-separated subset of the domains
    for name, d in fx["domains"].items():
        if only and name not in only.split(","):
            continue
        n_users = d.get("n_users", fx["n_users"])
        assert n_users == fx["n_users"] == 4096, "r04: one user count for all eight domains (ADVICE r03)"
        ar = torch.arange(n_users)
        dom = make_domain(name, d["n_items"], n_users, 32, cfg.vocab, d["seed"])
        seqs, at = [], 0
        for b in dom.sequence_batches:  # the fixture's labels (the reference's rank-derived items) replace the generator's random ones
            n = b.labels.numel()
            seqs.append(BatchSequence(sequence=b.sequence, labels=d["labels"][at:at + n].clone()))
            at += n
        _, metrics, scores, labels = test_model_on_dataloaders(module, [dom.item_batches], [seqs], [name], precision="bf16-mixed",
                                                               predictions_path=tmp_path / f"{name}.pt")
        assert model._weights.mode == "f16x3"
        got, E, U = scores[0], module.item_embeddings.detach().cpu(), module.eval_user_embeddings.detach().cpu()
        assert got.shape == (n_users, d["n_items"]) and torch.equal(labels[0], d["labels"])
        # (1) embeddings, catalog checksum, logits on the sampled columns
        rows, nu = d["E_rows"].long(), d["U"].shape[0]   # the fixture keeps the first users' embeddings and sampled catalog rows
        assert float((U[:nu] - d["U"]).abs().max()) < LOGIT_TOL and float((E[rows] - d["E_sample"]).abs().max()) < LOGIT_TOL
        assert abs(float(E.double().sum()) - d["E_checksum"]) < 1e-4 * d["n_items"]
        ref_idx, ref_val = d["ref_top52_idx"].long(), d["ref_top52_val"]
        logit_err = max(float((got[:nu][:, rows] - d["U"] @ d["E_sample"].T).abs().max()), float((got.gather(1, ref_idx) - ref_val).abs().max()),
                        float((got[ar, labels[0]] - d["label_score"]).abs().max()))
        assert logit_err < LOGIT_BOUND < LOGIT_TOL, (name, logit_err)
        NEAR_TIE = 2 * logit_err
        # (2) ranked indices: the reference's top-50, except where the reference's own scores are within NEAR_TIE of each other
        idx = module.eval_topk_indices.cpu()
        diff = idx != ref_idx[:, :50]
        for u, p in torch.nonzero(diff).tolist():
            hit = torch.nonzero(ref_idx[u] == idx[u, p]).flatten()
            assert hit.numel() == 1, (name, u, p, "an item outside the reference's top-52 entered the top-50")
            assert abs(float(ref_val[u, int(hit)] - ref_val[u, p])) <= NEAR_TIE, (name, u, p)
        gap = ref_val[:, :50] - ref_val[:, 1:51]
        above = torch.cat([torch.full_like(gap[:, :1], float("inf")), gap[:, :-1]], dim=1)
        clear = (gap > NEAR_TIE) & (above > NEAR_TIE)   # separated from both neighbours by more than twice the logit error
        assert bool((idx[clear] == ref_idx[:, :50][clear]).all()), (name, "a clearly separated rank position holds a different item")
        # (3) label ranks: a label may move only across items the reference scores within NEAR_TIE of it
        my_rank = (got > got[ar, labels[0]][:, None]).sum(1)
        ref_rank = d["label_rank"].long()
        half = (d["label_window"].shape[1] - 1) // 2  # reference scores kept either side of the label's rank
        assert half == 8, "every domain entry written by the committed generator revision (17-column label windows)"
        for u in torch.nonzero(my_rank != ref_rank).flatten().tolist():
            shift = int(my_rank[u] - ref_rank[u])
            assert abs(shift) <= half, (name, u, shift)
            lo, hi = sorted((half, half + shift))
            between = d["label_window"][u, lo:hi + 1]
            assert float((between - d["label_score"][u]).abs().max()) <= NEAR_TIE, (name, u, shift)
        # (4) metrics and loss: within 1e-3 of the reference, and EXACTLY the reference's values after the near-tie label moves verified above
        # (positions as the evaluators see them: the label's index in the ranked top-50 list, 50 = absent -- evaluator/metrics.py:51-57,79-86)
        pos = lambda lists: torch.where((lists == labels[0][:, None]).any(1), (lists == labels[0][:, None]).float().argmax(1), torch.full((n_users,), 50))
        ties = (d["label_window"][:, half - 1] == d["label_score"]) | (d["label_window"][:, half + 1] == d["label_score"])  # exact ties in the reference
        must, slack = O.metrics_after_rank_moves(d["metrics"], pos(ref_idx[:, :50]), pos(idx), fx["ks"], tie_users=ties)
        dom_worst = max((abs(metrics[0][k] - v), k) for k, v in d["metrics"].items())
        for k, v in d["metrics"].items():
            worst["any_metric"] = max(worst["any_metric"], abs(metrics[0][k] - v))
            assert abs(metrics[0][k] - must[k]) < 5e-6 + slack[k], (name, k, metrics[0][k], must[k], slack[k])
            assert abs(metrics[0][k] - v) <= NDCG_TOL, (name, k, metrics[0][k], v)
        assert abs(metrics[0]["test/loss"] - d["loss"]) < 1e-3, (name, metrics[0]["test/loss"], d["loss"])
        dn = abs(metrics[0]["test/NDCG@10"] - d["metrics"]["test/NDCG@10"])
        worst["logit"], worst["ndcg10"] = max(worst["logit"], logit_err), max(worst["ndcg10"], dn)
        print(f"[{name}] M={d['n_items']} logit max err {logit_err:.2e}; top-50 positions differing (all near-ties) {int(diff.sum())}; "
              f"labels moved {int((my_rank != ref_rank).sum())}; NDCG@10 {metrics[0]['test/NDCG@10']:.4f} (reference {d['metrics']['test/NDCG@10']:.4f}, |d| {dn:.1e}); "
              f"largest metric difference {dom_worst[0]:.1e} ({dom_worst[1]})")
    print(f"[8 domains] worst logit err {worst['logit']:.2e}, worst |dNDCG@10| {worst['ndcg10']:.1e}, worst |d metric| {worst['any_metric']:.1e}")
