"""north_star's accuracy clause for the second model family at true dimensions: BASELINE configs[2] -- a 3-domain task-vector merge of
Recformer-base (Longformer-base geometry, 148 M parameters; windowed + global attention, token-type and item-position embeddings) --
evaluated on a Pantry-sized domain (4,968 items, 1,024 users with sequences up to 1,024 tokens); and the same for Recformer-LARGE
(24 x 1,024, 435 M parameters: BASELINE configs[4]'s model) on 2,048 items and 512 users (fixture g15), and for BLaIR-LARGE (RoBERTa-large
geometry, 355 M parameters: `--model_type blair_large` of the reference's scripts; 2-domain merge; fixture g16).

Fixture: tests/golden/g14_realscale_recformer_base.pt, produced in the build container by oracle/gen_golden_recformer_realscale.py from the
reference itself (its load_merging_module / get_state_dict, its RecformerModel driving transformers' LongformerEncoder, user @ item.T, its
Evaluator; CPU, fp32).  Inputs are regenerated from seeds here.  Checked through the drop-in evaluation loop: embeddings and sampled logits
within 1e-4, the ranked top-50 equal up to the reference's own near-ties (2e-6), label ranks equal up to near-ties, every metric within 1e-3."""

import pytest
import torch

from oracle import ref_cpu as O
from tests.conftest import heavy, load_golden, prefetched, register_prefetch, seeded_state_dicts

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOGIT_TOL = 1e-4       # north_star
NEAR_TIE = 2e-6        # two items whose REFERENCE scores are this close may swap places (fp32 summation order)
NDCG_TOL = 1e-3        # north_star


SIZES = {"recformer_base": ("g14_realscale_recformer_base.pt", "RECFORMER_BASE"), "recformer_large": ("g15_realscale_recformer_large.pt", "RECFORMER_LARGE"),
         "blair_large": ("g16_realscale_blair_large.pt", "BLAIR_LARGE")}   # (BLaIR-base: tests/test_realscale_gpu.py, tests/test_8domain_gpu.py)


def _build_state_dicts(size):
    """host-only: the fixture and the pretrained / fine-tuned state dicts it names by seed (the reference model's key order: perturbations
    are drawn along it)"""
    fixture_name, model_type = SIZES[size]
    fx = load_golden(fixture_name)
    enc = fx.get("encoder", {})
    rec = model_type.startswith("RECFORMER")
    cfg = O.EncoderConfig(max_pos=4098, token_type_size=4, max_item_embeddings=51, one_sided_window=32, **enc) if rec else O.EncoderConfig(**enc)
    pre, fts = seeded_state_dicts(O.recformer_param_shapes(cfg) if rec else O.roberta_param_shapes(cfg), fx["key_order"], fx["seed_pre"], 0.02,
                                  fx["pre_checksum"], fx["seed_ft"], fx["ft_std"])
    return fx, cfg, rec, pre, fts


for _size in SIZES:
    register_prefetch(f"realscale:{_size}", (lambda s=_size: _build_state_dicts(s)), match=("test_realscale_recformer_gpu.py", f"[{_size}-"))


@pytest.fixture(scope="module", params=["recformer_base", pytest.param("recformer_large", marks=heavy), pytest.param("blair_large", marks=heavy)])
def setup(request):
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.model_batch import BatchSequence
    from mergerec_amd.module import ModelType
    from mergerec_amd.synthetic import make_domain

    model_type = SIZES[request.param][1]
    fx, cfg, rec, pre, fts = prefetched(f"realscale:{request.param}")  # host-only part, drawn in the background (tests/_prefetch.py)
    fsum = lambda sd: float(sum(v.double().sum() for v in sd.values() if v.is_floating_point()))
    model = ModelType[model_type].value(model_kwargs={"init_seed": 0, "device": DEV})
    model.load_state_dict(pre)
    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, model, pre, fts, set(), disable_softmax=True)  # merge_test.py:35-71
    mm.load_weights_from_dict({"global_weights": {"all": [1.0]}, "global_biases": {"all": [0.0]}, "per_weights": {"all": list(fx["alphas"])}})
    sd = {k: v.detach().clone() for k, v in mm.get_state_dict().items()}
    assert abs(fsum(sd) - fx["merged_checksum"]) < 1e-9 * max(1.0, abs(fx["merged_checksum"])) + 1e-5, (fsum(sd), fx["merged_checksum"])
    del mm, model, fts
    torch.cuda.empty_cache()
    dom = make_domain("Pantry", fx["n_items"], fx["n_users"], 32, cfg.vocab, fx["seed_domain"], kind="recformer" if rec else "roberta",
                      max_seq_len=fx.get("max_seq_len", 512))
    seqs, at = [], 0
    for b in dom.sequence_batches:  # the fixture's labels (the reference's rank-derived items) replace the generator's random ones
        n = b.labels.numel()
        seqs.append(BatchSequence(sequence=b.sequence, labels=fx["labels"][at:at + n].clone()))
        at += n
    if "longest_sequence" in fx:
        assert int(torch.cat([b.sequence["attention_mask"].sum(1) for b in seqs]).max()) == fx["longest_sequence"]
    return fx, sd, dom.item_batches, seqs, model_type


@pytest.mark.parametrize("mode,precision", [("f16x3", "bf16-mixed"), ("f32", "32-true")])
def test_logits_ranks_and_ndcg_match_the_reference(setup, mode, precision, tmp_path):
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.utils import test_model_on_dataloaders

    fx, sd, item_batches, seq_batches, model_type = setup
    model = ModelType[model_type].value(model_kwargs={"init_seed": 0, "device": DEV, "gemm_mode": mode})
    model.load_state_dict(sd)  # merge_test.py:71-80
    module = RecModule(model=model, evaluator=Evaluator(["NDCG", "RECALL"], fx["ks"]), similarity="cosine")
    _, metrics, scores, labels = test_model_on_dataloaders(module, [item_batches], [seq_batches], ["Pantry"], precision=precision,
                                                           predictions_path=tmp_path / "p.pt")
    assert model._weights.mode == mode
    n_users, M = fx["n_users"], fx["n_items"]
    got, E, U = scores[0], module.item_embeddings.detach().cpu(), module.eval_user_embeddings.detach().cpu()
    assert got.shape == (n_users, M) and torch.equal(labels[0], fx["labels"])
    rows, nu = fx["E_rows"].long(), fx["U"].shape[0]   # the fixture keeps the first users' embeddings and sampled catalog rows
    assert float((U[:nu] - fx["U"]).abs().max()) < LOGIT_TOL and float((E[rows] - fx["E_sample"]).abs().max()) < LOGIT_TOL
    assert abs(float(E.double().sum()) - fx["E_checksum"]) < 1e-4 * M
    logit_err = float((got[:nu][:, rows] - fx["U"] @ fx["E_sample"].T).abs().max())
    assert logit_err < LOGIT_TOL, logit_err
    idx = module.eval_topk_indices.cpu()
    ref_idx, ref_val = fx["ref_top52_idx"].long(), fx["ref_top52_val"]
    diff = idx != ref_idx[:, :50]
    for u, p in torch.nonzero(diff).tolist():
        hit = torch.nonzero(ref_idx[u] == idx[u, p]).flatten()
        assert hit.numel() == 1, (u, p, "an item outside the reference's top-52 entered the top-50")
        assert abs(float(ref_val[u, int(hit)] - ref_val[u, p])) <= NEAR_TIE, (u, p)
    gap = ref_val[:, :50] - ref_val[:, 1:51]
    above = torch.cat([torch.full_like(gap[:, :1], float("inf")), gap[:, :-1]], dim=1)
    clear = (gap > 2 * NEAR_TIE) & (above > 2 * NEAR_TIE)
    assert bool((idx[clear] == ref_idx[:, :50][clear]).all()), "a clearly separated rank position holds a different item"
    ar = torch.arange(n_users)
    my_rank = (got > got[ar, labels[0]][:, None]).sum(1)
    ref_rank = fx["label_rank"].long()
    for u in torch.nonzero(my_rank != ref_rank).flatten().tolist():
        shift = int(my_rank[u] - ref_rank[u])
        assert abs(shift) <= 3, (u, shift)
        lo, hi = sorted((3, 3 + shift))
        assert float((fx["label_window"][u, lo:hi + 1] - fx["label_score"][u]).abs().max()) <= NEAR_TIE, (u, shift)
    # the reference's values after the verified near-tie moves (positions as the evaluators see them: the label's index in the ranked
    # top-50 list, 50 = absent -- evaluator/metrics.py:51-57,79-86)
    pos = lambda lists: torch.where((lists == labels[0][:, None]).any(1), (lists == labels[0][:, None]).float().argmax(1), torch.full((n_users,), 50))
    ties = (fx["label_window"][:, 2] == fx["label_score"]) | (fx["label_window"][:, 4] == fx["label_score"])  # exact ties in the reference
    must, slack = O.metrics_after_rank_moves(fx["metrics"], pos(ref_idx[:, :50]), pos(idx), fx["ks"], tie_users=ties)
    for k, v in fx["metrics"].items():
        assert abs(metrics[0][k] - must[k]) < 5e-6 + slack[k], (k, metrics[0][k], must[k], slack[k])
        assert abs(metrics[0][k] - v) <= NDCG_TOL, (k, metrics[0][k], v)
    assert abs(metrics[0]["test/loss"] - fx["loss"]) < 1e-3
    print(f"[{model_type.lower()} {mode}] logit max err {logit_err:.2e}; top-50 positions differing (all near-ties) {int(diff.sum())}; labels moved "
          f"{int((my_rank != ref_rank).sum())}; NDCG@10 {metrics[0]['test/NDCG@10']:.4f} (reference {fx['metrics']['test/NDCG@10']:.4f})")
