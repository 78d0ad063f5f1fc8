"""Size-independent properties of the path AT BASELINE.json's full sizes (8-domain BLaIR-base: P = 124.6 M parameters, Arts-sized
catalog M = 22,855, 256 users + 128 items per step ~ 69 k tokens) -- where the CPU oracle would need minutes, the domain's own
invariants check the kernels: exact identities of the merge, slice / whole equivalence, order statistics of the top-k, batch-
composition invariance of the encoder, fixed points of the optimizer.  Everything goes through the C ABI (mergerec_amd.ops)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def arena():
    """BLaIR-base arena layout with seeded synthetic weights generated on the device (as bench.py does)."""
    from mergerec_amd.engine import ArenaLayout, EncoderSpec

    spec = EncoderSpec.blair_base()
    layout = ArenaLayout(spec.param_shapes("model."))
    P = layout.padded_numel
    g = torch.Generator(device=DEV).manual_seed(1234)
    base = torch.zeros(P, device=DEV)
    for k, shp in layout.shapes.items():
        v = layout.view(base, k)
        v.copy_(torch.randn(shp, generator=g, device=DEV) * 0.02)
        if "LayerNorm.weight" in k:
            v.add_(1.0)
    tv = torch.zeros(8, P, device=DEV)
    for i in range(8):
        for k, shp in layout.shapes.items():
            layout.view(tv[i], k).copy_(torch.randn(shp, generator=g, device=DEV) * 1e-3)
    return spec, layout, base, tv


def test_merge_exact_identities_at_full_size(arena):
    """alpha = 0 returns the base bit for bit; alpha = e_i returns base + tau_i (one rounding, = torch's elementwise add); merging the
    arena in 8 rank slices or in one piece gives the same bits; the alpha-gradient of g = tau_i is (<tau_j, tau_i>)_j"""
    from mergerec_amd import ops
    from mergerec_amd.parallel import SlicePlan

    spec, layout, base, tv = arena
    P, N = base.numel(), tv.shape[0]
    zero = torch.zeros(1, N, device=DEV)
    assert torch.equal(ops.merge_nway(base, tv, zero), base)
    for i in (0, 5):
        e = torch.zeros(1, N, device=DEV)
        e[0, i] = 1.0
        assert torch.equal(ops.merge_nway(base, tv, e), base + tv[i])
    alpha = torch.full((1, N), 1.0 / N, device=DEV)
    whole = ops.merge_nway(base, tv, alpha)
    plan = SlicePlan(P, 8)
    sliced = torch.full((plan.padded,), float("nan"), device=DEV)
    for r in range(8):
        lo, hi = plan.bounds(r)
        hi = min(hi, P)
        if hi > lo:
            ops.merge_nway(base, tv, alpha, None, out=sliced, p_begin=lo, p_count=hi - lo)
    assert torch.equal(sliced[:P], whole)
    # |merged - base| <= sum_i |alpha_i tau_i| (+ rounding): the merged model stays inside the task vectors' envelope
    assert bool(((whole - base).abs() <= (tv.abs().sum(0) / N) * (1 + 1e-5) + 2e-7 * (base.abs() + 1e-3)).all())
    g = ops.merge_bwd_alpha(tv, tv[3].contiguous())
    want = (tv.double() @ tv[3].double()).float()
    torch.testing.assert_close(g.view(-1), want, rtol=2e-5, atol=0)


def test_full_catalog_topk_order_statistics():
    """256 users x 22,855 items: values sorted, indices unique and pointing at their values, every score outside the list <= the last
    one inside, label rank = number of strictly greater scores (ties by index), log-sum-exp against torch"""
    from mergerec_amd import ops

    g = torch.Generator(device=DEV).manual_seed(5)
    nU, M, d, k = 256, 22855, 768, 50
    U = torch.nn.functional.normalize(torch.randn(nU, d, generator=g, device=DEV), dim=-1)
    E = torch.nn.functional.normalize(torch.randn(M, d, generator=g, device=DEV), dim=-1)
    E[M - 1] = E[7]  # an exact tie in every row
    labels = torch.randint(0, M, (nU,), generator=g, device=DEV)
    val, idx, lse, lab, rank, scores = ops.score_topk(U, E, k, labels, 20.0, return_scores=True)
    assert bool((val[:, :-1] >= val[:, 1:]).all())
    assert bool((torch.gather(scores, 1, idx) == val).all())
    assert all(len(set(r)) == k for r in idx.tolist()[:32])
    rest = scores.clone()
    rest.scatter_(1, idx, float("-inf"))
    assert bool((rest.max(dim=1).values <= val[:, -1]).all())
    tv_, _ = torch.topk(scores, k, dim=1)
    assert torch.equal(tv_, val)  # same multiset of values as torch.topk (order of exact ties aside)
    # canonical tie order: equal values appear with ascending index
    tie = val[:, :-1] == val[:, 1:]
    assert bool((idx[:, :-1][tie] < idx[:, 1:][tie]).all())
    lab_score = torch.gather(scores, 1, labels.view(-1, 1))
    greater = (scores > lab_score).sum(1) + ((scores == lab_score) & (torch.arange(M, device=DEV).view(1, M) < labels.view(-1, 1))).sum(1)
    want_rank = torch.where(greater < k, greater, torch.full_like(greater, -1)).to(torch.int32)
    assert torch.equal(rank, want_rank)
    torch.testing.assert_close(lse, torch.logsumexp(scores * 20.0, dim=1), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(lab, lab_score.view(-1) * 20.0, rtol=1e-6, atol=0)


@pytest.mark.parametrize("mode", ["f16x3", "bf16x3", "f32"])
def test_encoder_is_invariant_to_batch_composition_at_full_size(arena, mode):
    """~69 k packed tokens (256 users + 128 items, Amazon-shaped lengths): a sequence's embedding does not depend on what else is in
    the batch, on its position in it, or on the padding it arrived with -- bit for bit"""
    from mergerec_amd.engine import EncoderRunner, WeightSet
    from mergerec_amd.synthetic import _ids_from_lengths, blair_item_lengths, blair_sequence_lengths

    spec, layout, base, tv = arena
    W = WeightSet(layout, base, mode).refresh()
    run = EncoderRunner(spec)
    g = torch.Generator().manual_seed(3)
    lens = torch.cat([blair_item_lengths(128, g), blair_sequence_lengths(256, g)])
    batch = _ids_from_lengths(lens, spec.vocab, g)
    full = run.encode(W, batch, DEV, normalize=True, lens=lens)
    assert full.shape == (384, spec.hidden) and bool(torch.isfinite(full).all())
    torch.testing.assert_close(full.norm(dim=-1), torch.ones(384, device=DEV), rtol=0, atol=1e-5)
    # a subset, reversed, re-padded to its own maximum
    pick = torch.arange(383, 0, -7)
    L = int(lens[pick].max())
    sub = {k: v[pick][:, :L] for k, v in batch.items()}
    part = run.encode(W, sub, DEV, normalize=True, lens=lens[pick])
    assert torch.equal(part, full[pick.to(DEV)])
    # extra right padding changes nothing (on the sequences short enough to leave room in the 514-entry position table)
    short = pick[lens[pick] <= 400]
    Ls = int(lens[short].max())
    tight = {k: v[short][:, :Ls] for k, v in batch.items()}
    wide = {"input_ids": torch.nn.functional.pad(tight["input_ids"], (0, 9), value=spec.pad_id),
            "attention_mask": torch.nn.functional.pad(tight["attention_mask"], (0, 9), value=0)}
    assert torch.equal(run.encode(W, wide, DEV, normalize=True), full[short.to(DEV)])


def test_adamw_fixed_points_at_full_arena(arena):
    """lr = 0: parameters untouched, moments take (1 - beta) g and (1 - beta2) g^2; zero gradient + no decay: nothing moves; decay only:
    every decayed segment shrinks by exactly (1 - lr wd), biases / LayerNorm weights do not"""
    from mergerec_amd.optim import ArenaAdamW

    spec, layout, base, tv = arena
    p = base.clone()
    opt = ArenaAdamW(p, layout, lr=0.0, weight_decay=0.1)
    g = tv[1].contiguous()
    opt.step(g)
    assert torch.equal(p, base)
    torch.testing.assert_close(opt.exp_avg, (g * 0.1), rtol=1e-6, atol=0)
    torch.testing.assert_close(opt.exp_avg_sq, (g * g) * 0.001, rtol=1e-5, atol=0)
    p = base.clone()
    opt = ArenaAdamW(p, layout, lr=1e-2, weight_decay=0.0)
    opt.step(torch.zeros_like(p))
    assert torch.equal(p, base)
    opt = ArenaAdamW(p, layout, lr=1e-2, weight_decay=0.1)
    opt.step(torch.zeros_like(p))
    got, ref = layout.views(p), layout.views(base)
    for k in ("model.encoder.layer.3.output.dense.weight", "model.embeddings.word_embeddings.weight"):
        assert torch.equal(got[k], ref[k] * float(torch.tensor(1.0 - 1e-2 * 0.1, dtype=torch.float64).float()))
    for k in ("model.encoder.layer.3.output.dense.bias", "model.encoder.layer.7.attention.output.LayerNorm.weight"):
        assert torch.equal(got[k], ref[k])
