"""Next-row 2 (distillation losses, teacher matrix, alpha file) on the GPU, against the reference's own outputs
(tests/golden/g7_distill_losses.pt) and the CPU oracle."""
import pytest
import torch

from oracle import ref_cpu as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOSS_RTOL, GRAD_RTOL = 2e-5, 2e-4  # fp32 sums in a different order than torch's; exp/log of the device vs the host libm


def _make(name, T, coef, margin):
    from mergerec_amd.merger.enums import LossType
    from mergerec_amd.module import loss_fn as L

    if name == "PAIRWISE":
        return L.DistillPairwiseLoss(margin)
    if name == "LISTNET":
        return L.DistillListNetLoss(T)
    kw = {}
    if name.endswith("_KD"):
        kw["coefficient"] = coef
    return L.distill_loss_factory(LossType[name], temperature=T, **kw)


def test_losses_and_gradients_match_reference_golden():
    g7 = load_golden("g7_distill_losses.pt")
    seen = set()
    for c in g7["cases"]:
        fn = _make(c["loss"], c["temperature"], c["coefficient"], c["margin"])
        z = c["z"].to(DEV).requires_grad_(True)
        loss = fn(z, c["t"].to(DEV))
        loss.backward()
        ref, gref = c["value"], c["grad"]
        assert abs(loss.item() - ref.item()) <= LOSS_RTOL * max(1.0, abs(ref.item())), (c["loss"], tuple(c["z"].shape), loss.item(), ref.item())
        scale = max(gref.abs().max().item(), 1e-12)
        err = (z.grad.cpu() - gref).abs().max().item()
        assert err <= GRAD_RTOL * scale, (c["loss"], tuple(c["z"].shape), err, scale)
        seen.add(c["loss"])
    assert seen == set(O.DISTILL_LOSSES)


def test_row_losses_without_gradient_and_errors():
    from mergerec_amd import ops
    from mergerec_amd._lib import MergeRecHipError

    z = torch.randn(5, 77, device=DEV)
    t = torch.randn(5, 77, device=DEV)
    rows, dz = ops.distill_loss_rows(z, t, label_src=1, w_ce=1.0)
    assert dz is None
    want = torch.nn.functional.cross_entropy(z.cpu(), t.cpu().argmax(-1), reduction="none")
    assert torch.allclose(rows.cpu(), want, rtol=1e-5, atol=1e-6)
    with pytest.raises(MergeRecHipError):
        ops.distill_loss_rows(z, None, label_src=1, w_ce=1.0)       # teacher needed but missing
    with pytest.raises(MergeRecHipError):
        ops.distill_loss_rows(z, t, w_kd=1.0, temperature=0.0)      # temperature must be positive
    with pytest.raises(MergeRecHipError):
        ops.distill_loss_rows(z, t, label_src=0, w_ce=1.0)          # CE without a label source


def test_argmax_ties_go_to_lowest_index():
    from mergerec_amd import ops

    z = torch.zeros(2, 300, device=DEV)
    t = torch.zeros(2, 300, device=DEV)
    t[0, [7, 250]] = 1.0  # tie: label 7
    t[1, [299, 3]] = 2.0  # tie: label 3
    z[0, 7] = 5.0
    z[1, 3] = 5.0
    rows, _ = ops.distill_loss_rows(z, t, label_src=1, w_ce=1.0)
    want = torch.nn.functional.cross_entropy(z.cpu(), torch.tensor([7, 3]), reduction="none")
    assert torch.allclose(rows.cpu(), want, rtol=1e-5)


def test_teacher_scores_match_oracle_fma_chain():
    from mergerec_amd.module import teacher_scores
    from oracle import c_oracle

    g = torch.Generator().manual_seed(5)
    seq, item = torch.randn(37, 64, generator=g), torch.randn(211, 64, generator=g)
    got = teacher_scores(seq.to(DEV), item.to(DEV)).cpu()
    # same normalisation ops on the device (a row norm is a reduction: its rounding depends on the summation order), then the
    # product must be the ascending-k FMA chain bit for bit
    sd, idv = seq.to(DEV), item.to(DEV)
    sn = (sd / sd.norm(dim=-1, keepdim=True)).cpu()
    it = (idv / idv.norm(dim=-1, keepdim=True)).cpu()
    want = c_oracle.gemm_nt(sn.contiguous(), it.contiguous())
    assert torch.equal(got, want)
    assert torch.allclose(got, O.teacher_scores(seq, item), atol=2e-6)


class _FakeMerged(torch.nn.Module):
    """stands in for the merging module: 'encodes' a batch by looking representations up (the encoder is tested elsewhere)"""

    def __init__(self, reps):
        super().__init__()
        self.reps = reps

    def forward(self, batch):
        return self.reps[batch]

    def serialize_weights(self):
        return {"global_weights": {"g": [1.0]}, "global_biases": {"g": [0.0]}, "weights": {"g": [0.25, 0.75]}}


@pytest.mark.parametrize("n,M,d", [(1, 1000, 768), (2, 22855, 768), (5, 4968, 1024), (8, 333, 128), (11, 777, 64)])
def test_skinny_scores_and_their_gradient(n, M, d):
    """a handful of representation rows against a whole catalog (the distillation step's ``rep @ E_ds.T`` per sample): the streaming kernels
    against float64, forward and backward, row counts above one launch's eight, widths that are no multiple of 256"""
    from mergerec_amd import ops

    g = torch.Generator().manual_seed(n * 7 + M + d)
    reps, E = torch.randn(n, d, generator=g), torch.nn.functional.normalize(torch.randn(M, d, generator=g), dim=-1)
    out = torch.full((n, M + 5), 7.5, device=DEV)
    ops.skinny_scores(reps.to(DEV), E.to(DEV), out=out[:, :M])
    want = reps.double() @ E.double().T
    assert float((out[:, :M].cpu().double() - want).abs().max()) <= 2e-6 * float(reps.abs().sum(1).max()) and bool((out[:, M:] == 7.5).all())
    dz = torch.randn(n, M + 3, generator=g)
    got = ops.skinny_scores_bwd(dz.to(DEV)[:, :M], E.to(DEV), scale=0.5).cpu()
    wantg = 0.5 * (dz[:, :M].double() @ E.double())
    assert float((got.double() - wantg).abs().max()) <= 1e-5 * float(wantg.abs().max())
    again = ops.skinny_scores_bwd(dz.to(DEV)[:, :M], E.to(DEV), scale=0.5).cpu()
    assert torch.equal(got, again)  # fixed summation order


@pytest.mark.parametrize("batched", ["1", "0"])
def test_distill_module_matches_reference_loop_and_rep_gradient(batched, monkeypatch):
    monkeypatch.setenv("MR_DISTILL_BATCHED", batched)  # "1": one loss launch over every row (default); "0": the per-dataset loop of r01-r03
    from mergerec_amd.model_batch import BatchDistillationSequence
    from mergerec_amd.module import DistillSequenceModule
    from mergerec_amd.module.loss_fn import SinglePseudoLabelKDLoss

    f = load_golden("g7_distill_losses.pt")["forward_distill"]
    reps = f["reps"].to(DEV).requires_grad_(True)
    mod = DistillSequenceModule(_FakeMerged(reps), f["score_embeddings"], SinglePseudoLabelKDLoss(f["temperature"], f["coefficient"]), "dot")
    mod.item_embeddings = f["item_embeddings"]
    batch = BatchDistillationSequence(dataset_indexes=f["dataset_indexes"], sequence_ids=f["sequence_ids"], sequence=torch.arange(16, device=DEV))
    loss = mod(batch)
    loss.backward()
    assert abs(loss.item() - f["value"].item()) <= 5e-5 * abs(f["value"].item()), (loss.item(), f["value"].item())
    scale = f["rep_grad"].abs().max().item()
    assert (reps.grad.cpu() - f["rep_grad"]).abs().max().item() <= 5e-4 * scale
    # validation path: no graph, same value
    mod.on_validation_epoch_start()
    v = mod.validation_step(batch, 0)
    mod.on_validation_epoch_end()
    assert abs(v.item() - loss.item()) <= 1e-6 * abs(loss.item())
    assert "val/average_loss_epoch" in mod.logged
    # ids outside their tables raise on the host (upstream: torch's IndexError at module.py:66,68) instead of reaching the row-gather kernel
    n_rows = [t.shape[0] for t in f["score_embeddings"]]
    bad_sid = list(f["sequence_ids"]) if not isinstance(f["sequence_ids"], torch.Tensor) else f["sequence_ids"].clone()
    bad_sid[3] = n_rows[int(f["dataset_indexes"][3])]
    with pytest.raises(IndexError, match="teacher-score"):
        mod(BatchDistillationSequence(dataset_indexes=f["dataset_indexes"], sequence_ids=bad_sid, sequence=torch.arange(16, device=DEV)))
    bad_ds = list(f["dataset_indexes"])
    bad_ds[0] = len(n_rows)
    with pytest.raises(IndexError, match="dataset index"):
        mod(BatchDistillationSequence(dataset_indexes=bad_ds, sequence_ids=f["sequence_ids"], sequence=torch.arange(16, device=DEV)))


def test_alpha_file_round_trip(tmp_path):
    from mergerec_amd.module import SaveWeightsCallback
    from mergerec_amd.utils import load_alpha_file

    class T:
        current_epoch, global_step = 0, 10

    class M:
        merged_model = _FakeMerged(None)

    cb = SaveWeightsCallback(version="v", save_dir=tmp_path, log_every_steps=5)
    for i in range(11):
        cb.on_train_batch_end(T, M, None, None, i)
    cb.on_train_epoch_end(T, M)
    cb.teardown(T, M, "fit")
    assert load_alpha_file(tmp_path / "v.jsonl", -1) == M.merged_model.serialize_weights()
    assert len((tmp_path / "v.jsonl").read_text().strip().splitlines()) == 3
