"""Encoder backward (csrc/backward.hip + engine_train.py) against torch autograd through the CPU oracle restatement of the
reference's RoBERTa forward (oracle/ref_cpu.py, itself pinned by the transformers golden g3_roberta.pt)."""
from collections import OrderedDict

import pytest
import torch

from oracle import ref_cpu as O
from tests.conftest import heavy, load_golden, prefetched, register_prefetch, seeded_state_dicts

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_backward_kernels_match_torch():
    from mergerec_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 100, generator=g)
    t = ops.transpose_pad(x.to(DEV)).cpu()
    assert t.shape == (100, 48) and torch.equal(t[:, :37], x.T) and torch.count_nonzero(t[:, 37:]) == 0
    # 64 x 64 float4 tiles when the widths allow 16-byte accesses, the 32 x 32 scalar form otherwise; output into a column slice
    for R, C, pad in ((1000, 768, 32), (602, 3072, 16), (37, 99, 16), (130, 4, 16), (5, 260, 32)):
        xr = torch.randn(R, C, generator=g)
        got = ops.transpose_pad(xr.to(DEV), pad=pad).cpu()
        Rp = (R + pad - 1) // pad * pad
        assert got.shape == (C, Rp) and torch.equal(got[:, :R], xr[:, :C].T) and torch.count_nonzero(got[:, R:]) == 0, (R, C)
        wide = torch.full((C, Rp + 8), 7.0, device=DEV)
        ops.transpose_pad(xr.to(DEV), out=wide[:, :Rp], pad=pad)
        assert torch.equal(wide[:, :R].cpu(), xr[:, :C].T) and bool((wide[:, Rp:] == 7.0).all()), (R, C)
    assert torch.allclose(ops.colsum(x.to(DEV)).cpu(), x.sum(0), atol=1e-5)
    u, dh = torch.randn(50, 64, generator=g) * 2, torch.randn(50, 64, generator=g)
    uu = u.clone().requires_grad_(True)
    torch.nn.functional.gelu(uu).backward(dh)
    assert torch.allclose(ops.gelu_bwd(u.to(DEV), dh.to(DEV)).cpu(), uu.grad, atol=1e-6)
    # LayerNorm
    xx = (torch.randn(33, 128, generator=g) * 3 + 1).requires_grad_(True)
    gam, bet = torch.randn(128, generator=g).requires_grad_(True), torch.randn(128, generator=g).requires_grad_(True)
    dy = torch.randn(33, 128, generator=g)
    torch.nn.functional.layer_norm(xx, (128,), gam, bet, 1e-5).backward(dy)
    dg, db = torch.empty(128, device=DEV), torch.empty(128, device=DEV)
    dx = ops.layernorm_bwd(xx.detach().to(DEV), dy.to(DEV), gam.detach().to(DEV), 1e-5, dg, db).cpu()
    assert torch.allclose(dx, xx.grad, atol=2e-5, rtol=1e-4)
    assert torch.allclose(dg.cpu(), gam.grad, atol=2e-5, rtol=1e-4) and torch.allclose(db.cpu(), bet.grad, atol=2e-5, rtol=1e-4)
    # token-sized (fine-tuning) shapes: several 256-row chunks through the two-stage reductions
    xt = torch.randn(1000, 100, generator=g)
    assert torch.allclose(ops.colsum(xt.to(DEV)).cpu(), xt.sum(0), atol=1e-4)
    assert torch.allclose(ops.rowsum(xt.to(DEV)[:, :77]).cpu(), xt[:, :77].sum(1), atol=1e-4)
    xx = (torch.randn(777, 128, generator=g) * 3 + 1).requires_grad_(True)
    dy = torch.randn(777, 128, generator=g)
    gam.grad = bet.grad = None
    torch.nn.functional.layer_norm(xx, (128,), gam, bet, 1e-5).backward(dy)
    dx = ops.layernorm_bwd(xx.detach().to(DEV), dy.to(DEV), gam.detach().to(DEV), 1e-5, dg, db).cpu()
    assert torch.allclose(dx, xx.grad, atol=2e-5, rtol=1e-4)
    assert torch.allclose(dg.cpu(), gam.grad, atol=2e-4, rtol=1e-4) and torch.allclose(db.cpu(), bet.grad, atol=2e-4, rtol=1e-4)
    # hidden sizes that are whole multiples of 256 take the register-resident rows kernel (each element read once, 16-byte loads);
    # 18,000 rows = 71 row chunks through the eight-lane chunk combine
    for d, T in ((768, 300), (1024, 257), (256, 18000), (512, 40)):
        xx = (torch.randn(T, d, generator=g) * 3 + 1).requires_grad_(True)
        gam, bet = torch.randn(d, generator=g).requires_grad_(True), torch.randn(d, generator=g).requires_grad_(True)
        dy = torch.randn(T, d, generator=g)
        torch.nn.functional.layer_norm(xx, (d,), gam, bet, 1e-5).backward(dy)
        dg, db = torch.empty(d, device=DEV), torch.empty(d, device=DEV)
        dx = ops.layernorm_bwd(xx.detach().to(DEV), dy.to(DEV), gam.detach().to(DEV), 1e-5, dg, db).cpu()
        assert torch.allclose(dx, xx.grad, atol=3e-5, rtol=1e-4), (d, T)
        tol = 2e-4 * max(1.0, (T / 777) ** 0.5)
        assert torch.allclose(dg.cpu(), gam.grad, atol=tol, rtol=1e-4) and torch.allclose(db.cpu(), bet.grad, atol=tol, rtol=1e-4), (d, T)
    # scatter-add with repeated rows
    tab = torch.zeros(10, 64, device=DEV)
    idx = torch.tensor([3, 3, 9, 0, 3], dtype=torch.int32)
    src = torch.randn(5, 64, generator=g)
    ops.scatter_add_rows(src.to(DEV), idx.to(DEV), tab)
    want = torch.zeros(10, 64).index_add_(0, idx.long(), src)
    assert torch.allclose(tab.cpu(), want, atol=1e-6)


@pytest.mark.parametrize("lens", [[5, 1, 40, 33], [300, 17]])
def test_attention_backward_matches_torch(lens):
    from mergerec_amd import ops

    H, g = 2, torch.Generator().manual_seed(11)
    T = sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    qkv = torch.randn(T, 3 * H * 64, generator=g)
    dctx = torch.randn(T, H * 64, generator=g)
    q = qkv.clone().requires_grad_(True)
    outs = []
    for b in range(len(lens)):
        s, e = int(cu[b]), int(cu[b + 1])
        Q, K, V = (q[s:e, i * H * 64:(i + 1) * H * 64].view(e - s, H, 64).transpose(0, 1) for i in range(3))
        P = torch.softmax(Q @ K.transpose(1, 2) * 0.125, dim=-1)
        outs.append((P @ V).transpose(0, 1).reshape(e - s, H * 64))
    ctx = torch.cat(outs)
    ctx.backward(dctx)
    ctx_dev = ops.attention(qkv.to(DEV), cu.to(DEV), len(lens), H, max(lens), products=0)
    assert torch.allclose(ctx_dev.cpu(), ctx.detach(), atol=2e-5)
    got = ops.attention_bwd(qkv.to(DEV), ctx_dev, dctx.to(DEV), cu.to(DEV), len(lens), H).cpu()
    assert torch.allclose(got, q.grad, atol=3e-5, rtol=1e-4), (got - q.grad).abs().max()
    # the work-list launch (only the (sequence, 128-row block) pairs that exist) gives the same bits, with and without dropout, full and banded
    work = {128: (lambda w, n: (w.to(DEV), n))(*ops.attn_work_plan(torch.tensor(lens), 128))}
    for window, p_drop in ((-1, 0.0), (-1, 0.1), (4, 0.0), (4, 0.1)):
        kw = dict(window=window, max_len=max(lens), drop_p=p_drop, drop_key=77)
        box = ops.attention_bwd(qkv.to(DEV), ctx_dev, dctx.to(DEV), cu.to(DEV), len(lens), H, **kw)
        lst = ops.attention_bwd(qkv.to(DEV), ctx_dev, dctx.to(DEV), cu.to(DEV), len(lens), H, work=work, **kw)
        assert torch.equal(box, lst), (window, p_drop)


def test_encoder_backward_matches_oracle_autograd():
    """d (sum of CLS rows * R) / d every parameter, tiny RoBERTa config with true head size"""
    from mergerec_amd.engine import ArenaLayout, EncoderRunner
    from mergerec_amd.engine_train import RobertaTrainGraph, encode_with_grad
    from tests.test_path_gpu import _spec

    g3 = load_golden("g3_roberta.pt")
    cfgd, sd = g3["cfg"], g3["state_dict"]
    cfg = O.EncoderConfig(**{k: cfgd[k] for k in cfgd if k in O.EncoderConfig.__dataclass_fields__})
    ids, mask = g3["input_ids"], g3["attention_mask"]
    # CPU: autograd through the oracle forward
    p = OrderedDict((k, v.clone().float().requires_grad_(v.is_floating_point())) for k, v in sd.items())
    cls = O.roberta_encode(p, ids, mask, cfg, prefix="model.")
    R = torch.randn(cls.shape, generator=torch.Generator().manual_seed(5))
    (O.maybe_normalize(cls) * R).sum().backward()
    # GPU
    views = OrderedDict((k, v.to(torch.float32)) for k, v in sd.items())
    layout = ArenaLayout(OrderedDict((k, tuple(v.shape)) for k, v in views.items()))
    flat = layout.pack(views, DEV).requires_grad_(True)
    spec = _spec(cfgd)
    pb = EncoderRunner(spec).pack({"input_ids": ids, "attention_mask": mask}, DEV)
    graph = RobertaTrainGraph(spec, layout)
    out = encode_with_grad(graph, flat, pb)
    assert torch.allclose(out.detach().cpu(), cls.detach(), atol=1e-4, rtol=1e-5)
    (torch.nn.functional.normalize(out, dim=-1) * R.to(DEV)).sum().backward()
    got = layout.views(flat.grad)
    worst = 0.0
    gmax = max(float(v.grad.abs().max()) for v in p.values() if v.requires_grad and v.grad is not None)
    for k, v in p.items():
        if not v.requires_grad or v.grad is None:
            assert float(got[k].abs().max()) == 0.0, k  # pooler, buffers: untouched by the CLS path
            continue
        ref = v.grad
        # (key biases have a mathematically zero gradient -- softmax is shift invariant -- so their scale is floored)
        scale = max(float(ref.abs().max()), 1e-3 * gmax)
        err = float((got[k].cpu() - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 2e-3, (k, err, scale)
    assert worst > 0.0


@pytest.mark.parametrize("learn,mode", [("TASK_WISE", "f32"), ("LAYER_WISE", "f32"), ("TASK_WISE", "bf16x3"), ("LAYER_WISE", "bf16x3")])
def test_alpha_gradient_end_to_end_matches_oracle_autograd(learn, mode):
    """alpha -> merge -> encoder -> cosine logits -> SINGLE_PSEUDO_LABEL_KD, all on the device, against torch autograd through
    the CPU oracle of every stage (the reference's merge_train.py step, module/distiller/sequence/module.py:59-79)"""
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.model_batch import BatchDistillationSequence
    from mergerec_amd.module import DistillSequenceModule
    from mergerec_amd.module.loss_fn import SinglePseudoLabelKDLoss
    from tests.test_path_gpu import _tiny_model

    g2 = load_golden("g2_merger.pt")
    cfgd = g2["cfg"]
    cfg = O.EncoderConfig(**{k: cfgd[k] for k in cfgd if k in O.EncoderConfig.__dataclass_fields__})
    ids, mask = g2["input_ids"], g2["attention_mask"]
    B = ids.shape[0]
    gen = torch.Generator().manual_seed(9)
    items = [torch.nn.functional.normalize(torch.randn(m, cfgd["hidden"], generator=gen), dim=-1) for m in (50, 77)]
    teachers = [torch.randn(B, m, generator=gen).clamp(-1, 1) for m in (50, 77)]
    ds_idx = [i % 2 for i in range(B)]
    seq_ids = list(range(B))
    T, COEF = 0.05, 1000.0

    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType[learn], _tiny_model(cfgd), g2["pretrain"], [dict(f) for f in g2["finetunes"]], set(),
                             disable_softmax=True, initial_per_weight=0.3)
    mm.train_mode = mode  # exact-fp32 products, or the split-precision graph merge_train.py uses under --precision bf16-mixed
    mod = DistillSequenceModule(mm, teachers, SinglePseudoLabelKDLoss(T, COEF), "cosine",
                                trainable_args_kwargs={"freeze_global_weight": True, "freeze_global_bias": True})
    mod.item_embeddings = items
    batch = BatchDistillationSequence(dataset_indexes=ds_idx, sequence_ids=torch.tensor(seq_ids), sequence={"input_ids": ids, "attention_mask": mask})
    mod.train()
    loss = mod.training_step(batch.to(DEV), 0)
    loss.backward()

    # ---- CPU: the same chain with torch autograd through the oracle
    pre, fts = O.align_state_dicts(g2["pretrain"], g2["finetunes"])
    base, shapes = O.flatten_model(pre)
    tv = O.get_task_vectors(base, [O.flatten_model(f)[0] for f in fts])
    n = tv.shape[0]
    if learn == "TASK_WISE":  # alpha = gw * per + gb with gw = 1, gb = 0 (task_wise.py:37-42)
        per = {"all": torch.full((n,), 0.3, requires_grad=True)}
        merged = O.merge_task_wise(base, tv, 1.0 * per["all"] + 0.0)
    else:
        groups = O.group_parameters_by_layer(shapes)
        per = {k: torch.full((n,), 0.3, requires_grad=True) for k in groups}
        merged = O.merge_layer_wise(base, tv, groups, {k: 1.0 * v + 0.0 for k, v in per.items()})
    sd = O.get_state_dict(merged, shapes)
    reps = O.maybe_normalize(O.roberta_encode(sd, ids, mask, cfg, prefix="model."))
    ref = O.forward_distill(reps, items, teachers, ds_idx, seq_ids, lambda z, t: O.distill_loss("SINGLE_PSEUDO_LABEL_KD", z, t, T, COEF))
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 2e-4 * abs(ref.item()), (loss.item(), ref.item())
    for k, p in per.items():
        got = mm.per_weights[k].grad.cpu()
        scale = max(float(p.grad.abs().max()), 1e-6)
        assert float((got - p.grad).abs().max()) <= 5e-3 * scale, (k, got, p.grad)


def test_bad_token_ids_raise_on_every_path_that_packs():
    """the packing kernel clamps a bad id and flags it; every loop that packs must read the flag.  The bare public
    forwards report at once (like nn.Embedding's IndexError upstream); the alpha-learning loop at its first step; a RecModule that wraps
    a merging module at the epoch end; a catalog encode right after the encode."""
    from mergerec_amd.engine import InputError, check_module_inputs
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.model_batch import BatchDistillationSequence, BatchItem, BatchSequence
    from mergerec_amd.module import DistillSequenceModule, RecModule
    from mergerec_amd.module.callbacks import ItemEncoderMixin
    from mergerec_amd.module.loss_fn import SinglePseudoLabelKDLoss
    from mergerec_amd.utils import DistillTrainer
    from tests.test_path_gpu import _tiny_model

    g2 = load_golden("g2_merger.pt")
    cfgd = g2["cfg"]
    ids, mask = g2["input_ids"], g2["attention_mask"]
    B = ids.shape[0]
    bad_ids = ids.clone()
    bad_ids[1, 2] = cfgd["vocab"] + 5
    good, bad = {"input_ids": ids, "attention_mask": mask}, {"input_ids": bad_ids, "attention_mask": mask}
    model = _tiny_model(cfgd)
    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, model, g2["pretrain"], [dict(f) for f in g2["finetunes"]], set(),
                             disable_softmax=True, initial_per_weight=0.3)
    dev = lambda b: {k: v.to(DEV) for k, v in b.items()}
    # bare forwards: at once
    with pytest.raises(InputError, match="input_ids"):
        model.forward(dev(bad))
    with torch.no_grad():
        with pytest.raises(InputError, match="input_ids"):
            mm.forward(dev(bad))
        mm.forward(dev(good))
    # the training graph under the merging module: deferred, visible through the module tree
    out = mm.forward_with_grad(dev(bad))
    assert torch.isfinite(out).all()
    with pytest.raises(InputError, match="input_ids"):
        check_module_inputs(mm)
    check_module_inputs(mm)  # cleared
    # a RecModule around the merging module (the hasattr guard of r02 skipped it)
    rec = RecModule(model=mm, evaluator=Evaluator(["NDCG"], [5]), similarity="cosine")
    rec.eval()
    gen = torch.Generator().manual_seed(3)
    rec.item_embeddings = torch.nn.Parameter(torch.nn.functional.normalize(torch.randn(40, cfgd["hidden"], generator=gen), dim=-1).to(DEV), requires_grad=False)
    rec.on_test_epoch_start()
    with torch.no_grad():
        rec.test_step(BatchSequence(sequence=bad, labels=torch.randint(0, 40, (B,), generator=gen)).to(DEV), 0)
        with pytest.raises(InputError, match="input_ids"):
            rec.on_test_epoch_end()
    # the catalog encode
    rec2 = RecModule(model=model, evaluator=Evaluator(["NDCG"], [5]), similarity="cosine")
    with pytest.raises(InputError, match="input_ids"):
        ItemEncoderMixin.encode_items([BatchItem(items=good), BatchItem(items=bad)], rec2)
    assert ItemEncoderMixin.encode_items([BatchItem(items=good)], rec2).shape == (B, cfgd["hidden"])

    # one alpha-learning step on a bad batch: the trainer stops at step 1 instead of optimising alpha on clamped embeddings
    items = [torch.nn.functional.normalize(torch.randn(m, cfgd["hidden"], generator=gen), dim=-1) for m in (50, 77)]
    teachers = [torch.randn(B, m, generator=gen).clamp(-1, 1) for m in (50, 77)]
    mod = DistillSequenceModule(mm, teachers, SinglePseudoLabelKDLoss(0.05, 1000.0), "cosine",
                                trainable_args_kwargs={"freeze_global_weight": True, "freeze_global_bias": True})
    mod.item_embeddings = items

    class OneBatch:
        def __init__(self, enc):
            self.enc = enc

        def setup(self, stage):
            pass

        def train_dataloader(self):
            return [BatchDistillationSequence(dataset_indexes=[i % 2 for i in range(B)], sequence_ids=torch.arange(B), sequence=self.enc)]

    with pytest.raises(InputError, match="input_ids"):
        DistillTrainer(max_steps=1, verbose=False).fit(mod, OneBatch(bad))
    hist = DistillTrainer(max_steps=1, verbose=False).fit(mod, OneBatch(good))
    assert len(hist) == 1 and hist[0] == hist[0]


def test_merge_train_cli_runs_and_moves_alpha(tmp_path):
    import sys

    root = __import__("pathlib").Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import merge_train
    from mergerec_amd.engine import EncoderSpec
    from mergerec_amd.module import models
    from mergerec_amd.utils import load_alpha_file
    from tests.conftest import GOLDEN

    old = models.BLaIRBase.SPEC
    models.BLaIRBase.SPEC = staticmethod(lambda: EncoderSpec(hidden=128, heads=2, layers=2, intermediate=256, vocab=50265, max_pos=514))
    try:
        res = merge_train.main([
            "--model_type", "blair_base", "--model_kwargs", "init_seed", "7", "--finetune_checkpoint_paths", "synthetic:1", "synthetic:2",
            "--data_paths", str(GOLDEN / "mini_dataset"), str(GOLDEN / "mini_dataset"), "--tokenizer_path", str(GOLDEN / "mini_tokenizer"),
            "--item_embeddings_paths", "auto", "--sequence_embeddings_paths", "auto", "--train_data_split", "item", "--test_data_split", "test",
            "--merge_type", "task_vector", "--learn_type", "task_wise", "--loss_type", "SINGLE_PSEUDO_LABEL_KD", "--coefficient", "1000",
            "--learning_rate", "0.01", "--max_steps", "12", "--batch_size", "16", "--max_seq_len", "96", "--max_attribute_len", "12", "--max_items", "20",
            "--weights_dir", str(tmp_path)])
    finally:
        models.BLaIRBase.SPEC = staticmethod(old)
    hist = res["history"]
    assert len(hist) == 12 and all(h == h and abs(h) < 1e6 for h in hist)
    per = res["weights"]["per_weights"]["all"]
    assert any(abs(w - 0.2) > 1e-3 for w in per), per          # Adam moved alpha away from initial_per_weight
    assert load_alpha_file(res["weights_file"], -1)["per_weights"]["all"] != [0.2, 0.2]
    assert "test/dataset_0/test/NDCG@10" in res["test_metrics"]


def test_global_row_backward_matches_torch():
    from mergerec_amd import ops

    H, lens, g = 2, [7, 300, 1], torch.Generator().manual_seed(21)
    B, T = len(lens), sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    qg = torch.randn(B, H * 64, generator=g).requires_grad_(True)
    kvg = torch.randn(T, 2 * H * 64, generator=g).requires_grad_(True)
    dctx = torch.randn(B, H * 64, generator=g)
    outs = []
    for b in range(B):
        s, e = int(cu[b]), int(cu[b + 1])
        q = qg[b].view(H, 1, 64)
        K = kvg[s:e, :H * 64].view(e - s, H, 64).transpose(0, 1)
        V = kvg[s:e, H * 64:].view(e - s, H, 64).transpose(0, 1)
        outs.append((torch.softmax(q @ K.transpose(1, 2) * 0.125, dim=-1) @ V).reshape(H * 64))
    ctx = torch.stack(outs)
    ctx.backward(dctx)
    full = torch.zeros(T, H * 64, device=DEV)
    ops.attention_global_row(qg.detach().to(DEV), kvg.detach().to(DEV), cu.to(DEV), B, H, max(lens), full)
    got_ctx = full[cu[:-1].long().to(DEV)]
    assert torch.allclose(got_ctx.cpu(), ctx.detach(), atol=2e-5)
    dq, dkv = ops.attention_global_row_bwd(qg.detach().to(DEV), kvg.detach().to(DEV), got_ctx.contiguous(), dctx.to(DEV), cu.to(DEV), B, H)
    assert torch.allclose(dq.cpu(), qg.grad, atol=3e-5, rtol=1e-4) and torch.allclose(dkv.cpu(), kvg.grad, atol=3e-5, rtol=1e-4)


def test_recformer_backward_matches_oracle_autograd():
    """every parameter gradient of the Longformer-style encoder (band + global key, global CLS row, four embedding tables)"""
    from mergerec_amd.engine import ArenaLayout, EncoderRunner
    from mergerec_amd.engine_train import EncoderTrainGraph, encode_with_grad
    from tests.test_path_gpu import _spec

    for case in load_golden("g4_recformer.pt")["cases"]:
        cfgd, sd, b = case["cfg"], case["state_dict"], case["batch"]
        cfg = O.EncoderConfig(**{k: cfgd[k] for k in cfgd if k in O.EncoderConfig.__dataclass_fields__})
        p = OrderedDict((k, v.clone().float().requires_grad_(v.is_floating_point())) for k, v in sd.items())
        cls = O.recformer_encode(p, b["input_ids"], b["attention_mask"], b["global_attention_mask"], b["token_type_ids"], b["item_position_ids"], cfg,
                                 prefix="model.")
        R = torch.randn(cls.shape, generator=torch.Generator().manual_seed(5))
        (O.maybe_normalize(cls) * R).sum().backward()
        views = OrderedDict((k, v.to(torch.float32)) for k, v in sd.items())
        layout = ArenaLayout(OrderedDict((k, tuple(v.shape)) for k, v in views.items()))
        flat = layout.pack(views, DEV).requires_grad_(True)
        spec = _spec(cfgd, "recformer")
        pb = EncoderRunner(spec).pack(b, DEV)
        out = encode_with_grad(EncoderTrainGraph(spec, layout), flat, pb)
        assert torch.allclose(out.detach().cpu(), cls.detach(), atol=1e-4, rtol=1e-5)
        (torch.nn.functional.normalize(out, dim=-1) * R.to(DEV)).sum().backward()
        got = layout.views(flat.grad)
        gmax = max(float(v.grad.abs().max()) for v in p.values() if v.requires_grad and v.grad is not None)
        for k, v in p.items():
            if not v.requires_grad or v.grad is None:
                assert float(got[k].abs().max()) == 0.0, k
                continue
            scale = max(float(v.grad.abs().max()), 1e-3 * gmax)
            err = float((got[k].cpu() - v.grad).abs().max()) / scale
            assert err <= 3e-3, (k, err, scale)


def test_recformer_backward_matches_reference_gradients():
    """g11: every parameter gradient of the reference's RecformerModel (its own embeddings / mask helpers driving the library's
    LongformerEncoder under autograd, recorded in the build container) against the HIP training graph -- no oracle in between"""
    from mergerec_amd.engine import ArenaLayout, EncoderRunner
    from mergerec_amd.engine_train import EncoderTrainGraph, encode_with_grad
    from tests.test_path_gpu import _spec

    for case, gr in zip(load_golden("g4_recformer.pt")["cases"], load_golden("g11_recformer_grads.pt")["cases"]):
        cfgd, sd, b = case["cfg"], case["state_dict"], case["batch"]
        views = OrderedDict((k, v.to(torch.float32)) for k, v in sd.items())
        layout = ArenaLayout(OrderedDict((k, tuple(v.shape)) for k, v in views.items()))
        flat = layout.pack(views, DEV).requires_grad_(True)
        spec = _spec(cfgd, "recformer")
        out = encode_with_grad(EncoderTrainGraph(spec, layout), flat, EncoderRunner(spec).pack(b, DEV))
        (torch.nn.functional.normalize(out, dim=-1) * gr["R"].to(DEV)).sum().backward()
        got = layout.views(flat.grad)
        gmax = max(float(g.abs().max()) for g in gr["grads"].values() if g is not None)
        for k, g in gr["grads"].items():
            if g is None:
                assert float(got[k].abs().max()) == 0.0, k
                continue
            scale = max(float(g.abs().max()), 1e-3 * gmax)
            err = float((got[k].cpu() - g).abs().max()) / scale
            assert err <= 3e-3, (k, err, scale)


def test_merge_train_cli_recformer(tmp_path):
    """the optimisation loop on a Recformer-shaped model: pre-tokenised item sequences, Longformer attention backward, layer-wise alpha"""
    import sys

    root = __import__("pathlib").Path(__file__).resolve().parent.parent
    sys.path.insert(0, str(root))
    import merge_train
    from mergerec_amd.engine import EncoderSpec
    from mergerec_amd.module import models
    from tests.conftest import GOLDEN

    old = models.RecformerBase.SPEC
    models.RecformerBase.SPEC = staticmethod(lambda: EncoderSpec(kind="recformer", hidden=128, heads=2, layers=2, intermediate=256, vocab=50265, max_pos=1026,
                                                                 token_type_size=4, max_item_embeddings=51, one_sided_window=32))
    try:
        res = merge_train.main([
            "--model_type", "recformer_base", "--model_kwargs", "init_seed", "7", "--finetune_checkpoint_paths", "synthetic:1", "synthetic:2",
            "--data_paths", str(GOLDEN / "mini_dataset"), str(GOLDEN / "mini_dataset"), "--tokenizer_path", str(GOLDEN / "mini_tokenizer"),
            "--item_embeddings_paths", "auto", "--sequence_embeddings_paths", "auto", "--train_data_split", "item", "--test_data_split", "test",
            "--merge_type", "ties", "--learn_type", "layer_wise", "--loss_type", "SINGLE_PSEUDO_LABEL_KD", "--coefficient", "1000",
            "--learning_rate", "0.01", "--max_epochs", "2", "--valid_ratio", "0.25", "--batch_size", "16", "--max_seq_len", "128",
            "--max_attribute_len", "10", "--max_items", "20", "--weights_dir", str(tmp_path)])
    finally:
        models.RecformerBase.SPEC = staticmethod(old)
    # 2 domains x 60 items, 25 % held out -> 90 training pseudo users -> 6 steps per epoch; validation after every epoch picks the alpha
    assert len(res["history"]) == 12 and all(h == h and abs(h) < 1e6 for h in res["history"])
    per = res["weights"]["per_weights"]
    assert set(per) >= {"others", "0", "1"} and any(abs(w - 0.2) > 1e-4 for ws in per.values() for w in ws), per
    assert "test/dataset_0/test/NDCG@10" in res["test_metrics"]


@pytest.mark.parametrize("case", [0, 1])
def test_whole_merge_train_step_matches_reference(case):
    """g10: one collaborative-merging step of the reference ITSELF (its load_merging_module + DistillSequenceModule.training_step around
    transformers' RobertaModel, recorded in the build container): loss and d loss / d (per_weights, global_weights, global_biases),
    task-wise and layer-wise, against the HIP step -- no oracle in between"""
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.model_batch import BatchDistillationSequence
    from mergerec_amd.module import DistillSequenceModule, ModelType
    from mergerec_amd.module.loss_fn import SinglePseudoLabelKDLoss

    g10 = load_golden("g10_merge_train_step.pt")
    c = g10["cases"][case]
    cfgd = g10["cfg"]
    fts = [O.perturbed_state_dict(g10["pretrain"], seed=s, std=g10["finetune_std"]) for s in g10["finetune_seeds"]]
    over = dict(hidden=cfgd["hidden"], heads=cfgd["heads"], layers=cfgd["layers"], intermediate=cfgd["intermediate"], vocab=cfgd["vocab"],
                max_pos=cfgd["max_pos"])
    model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 1, "spec_overrides": over, "device": DEV, "hidden_dropout_prob": 0.0, "attention_probs_dropout_prob": 0.0})
    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType[c["learn_type"]], model, g10["pretrain"], [dict(f) for f in fts], set(),
                             disable_softmax=True, initial_per_weight=g10["initial_per_weight"])
    assert list(mm.per_weights.keys()) == c["groups"]
    mod = DistillSequenceModule(mm, g10["score_embeddings"], SinglePseudoLabelKDLoss(g10["temperature"], g10["coefficient"]), "cosine")
    mod.item_embeddings = g10["item_embeddings"]
    batch = BatchDistillationSequence(dataset_indexes=g10["dataset_indexes"], sequence_ids=torch.tensor(g10["sequence_ids"]),
                                      sequence={"input_ids": g10["input_ids"], "attention_mask": g10["attention_mask"]})
    mod.train()
    loss = mod.training_step(batch.to(DEV), 0)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), c["loss"], rtol=5e-5, atol=5e-5)
    for name in ("per_weights", "global_weights", "global_biases"):
        for k in c["groups"]:
            want, got = c["grads"][name][k], getattr(mm, name)[k].grad.cpu()
            scale = max(float(want.abs().max()), 1e-3)
            assert float((got - want).abs().max()) <= 5e-3 * scale, (name, k, got, want)


def _build_g19():
    """host-only: the regenerated inputs of fixture g19 (pretrained + 8 fine-tuned state dicts, item matrices, teacher scores)"""
    fx = load_golden("g19_merge_train_step_realscale.pt")
    cfg = O.EncoderConfig()
    pre, fts = seeded_state_dicts(O.roberta_param_shapes(cfg), fx["key_order"], fx["pretrain_seed"], fx["pretrain_std"], fx["pretrain_checksum"],
                                  fx["finetune_seeds"], fx["finetune_std"], fx["finetune_checksum"])
    # the generator's data stream: lengths, ids, then the item matrices, then the teacher scores (one generator)
    g = torch.Generator().manual_seed(fx["data_seed"])
    B, L = fx["batch"]
    lens = torch.randint(3, L + 1, (B,), generator=g)
    torch.randint(3, cfg.vocab, (B, L), generator=g)
    assert torch.equal(((torch.arange(L).view(1, L) < lens.view(B, 1)).long()), fx["attention_mask"])
    items = [torch.nn.functional.normalize(torch.randn(m, cfg.hidden, generator=g), dim=-1) for m in fx["catalog_sizes"]]
    teachers = [torch.randn(B, m, generator=g).clamp(-1, 1) for m in fx["catalog_sizes"]]
    assert abs(float(sum(x.double().sum() for x in items)) - fx["item_checksum"]) < 1e-6 * abs(fx["item_checksum"]) + 1e-6
    assert abs(float(sum(x.double().sum() for x in teachers)) - fx["teacher_checksum"]) < 1e-6 * abs(fx["teacher_checksum"]) + 1e-6
    return fx, pre, fts, items, teachers


register_prefetch("g19", _build_g19, match=("test_train_gpu.py", "test_whole_merge_train_step_at_real_dimensions"))


@pytest.fixture(scope="module")
def g19_inputs():
    """built once for both cases (drawn in the background while earlier modules run: tests/_prefetch.py)"""
    return prefetched("g19")


@pytest.mark.parametrize("case", [0, 1])
def test_whole_merge_train_step_at_real_dimensions_matches_reference(g19_inputs, case):
    """g19 (oracle/gen_golden_merge_train_realscale.py): the reference's collaborative-merging step at BLaIR-base's TRUE dimensions with 8
    fine-tuned checkpoints (a 4 GB task-vector matrix under torch autograd on the CPU), 16 pseudo users, the eight real catalog sizes --
    loss and d loss / d (per_weights, global_weights, global_biases), task-wise and layer-wise (13 groups), against the HIP step."""
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.model_batch import BatchDistillationSequence
    from mergerec_amd.module import DistillSequenceModule, ModelType
    from mergerec_amd.module.loss_fn import SinglePseudoLabelKDLoss

    fx, pre, fts, items, teachers = g19_inputs
    c = fx["cases"][case]
    model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 1, "device": DEV, "hidden_dropout_prob": 0.0, "attention_probs_dropout_prob": 0.0})
    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType[c["learn_type"]], model, pre, [dict(f) for f in fts], set(),
                             disable_softmax=True, initial_per_weight=fx["initial_per_weight"])
    assert list(mm.per_weights.keys()) == c["groups"]
    mod = DistillSequenceModule(mm, teachers, SinglePseudoLabelKDLoss(fx["temperature"], fx["coefficient"]), "cosine")
    mod.item_embeddings = items
    batch = BatchDistillationSequence(dataset_indexes=fx["dataset_indexes"], sequence_ids=torch.tensor(fx["sequence_ids"]),
                                      sequence={"input_ids": fx["input_ids"], "attention_mask": fx["attention_mask"]})
    mod.train()
    loss = mod.training_step(batch.to(DEV), 0)
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), c["loss"], rtol=5e-5, atol=5e-5)
    worst = 0.0
    for name in ("per_weights", "global_weights", "global_biases"):
        for k in c["groups"]:
            want, got = c["grads"][name][k], getattr(mm, name)[k].grad.cpu()
            scale = max(float(want.abs().max()), 1e-3)
            worst = max(worst, float((got - want).abs().max()) / scale)
            assert float((got - want).abs().max()) <= 5e-3 * scale, (name, k, got, want)
    print(f"[merge_train step, BLaIR-base x 8 domains, {c['learn_type']}] loss {float(loss.detach()):.6f} (reference {float(c['loss']):.6f}); "
          f"worst gradient deviation {worst:.1e} of the group's largest entry")


def _build_g20():
    """host-only: the regenerated inputs of fixture g20 (Recformer-large: pretrained + 4 fine-tuned state dicts of 435 M parameters each,
    item matrices, teacher scores)"""
    fx = load_golden("g20_merge_train_step_recformer_large.pt")
    cfg = O.EncoderConfig(**{k: v for k, v in fx["cfg"].items() if k in O.EncoderConfig.__dataclass_fields__})
    pre, fts = seeded_state_dicts(O.recformer_param_shapes(cfg), fx["key_order"], fx["pretrain_seed"], fx["pretrain_std"], fx["pretrain_checksum"],
                                  fx["finetune_seeds"], fx["finetune_std"], fx["finetune_checksum"])
    g = torch.Generator().manual_seed(fx["data_seed"])
    B, L = fx["batch"]
    torch.randint(3, L + 1, (B,), generator=g)            # (the generator's stream: lengths, ids, items, teachers)
    torch.randint(3, cfg.vocab, (B, L), generator=g)
    items = [torch.nn.functional.normalize(torch.randn(m, cfg.hidden, generator=g), dim=-1) for m in fx["catalog_sizes"]]
    teachers = [torch.randn(B, m, generator=g).clamp(-1, 1) for m in fx["catalog_sizes"]]
    assert abs(float(sum(x.double().sum() for x in items)) - fx["item_checksum"]) < 1e-6 * abs(fx["item_checksum"]) + 1e-6
    assert abs(float(sum(x.double().sum() for x in teachers)) - fx["teacher_checksum"]) < 1e-6 * abs(fx["teacher_checksum"]) + 1e-6
    return fx, pre, fts, items, teachers


register_prefetch("g20", _build_g20, match=("test_train_gpu.py", "recformer_large"))


@pytest.fixture(scope="module")
def g20_inputs():
    """built ONCE for the four tests that use them (two reference cases, two 8-domain property cases), in the background while earlier
    modules run (tests/_prefetch.py)"""
    return prefetched("g20")


def _g20_step(g20_inputs, case, copies, per_weight):
    """one collaborative-merging step on g20's inputs with the fine-tuned checkpoints listed `copies` times -> (module, loss)"""
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.model_batch import BatchDistillationSequence
    from mergerec_amd.module import DistillSequenceModule, ModelType
    from mergerec_amd.module.loss_fn import SinglePseudoLabelKDLoss

    fx, pre, fts, items, teachers = g20_inputs
    c = fx["cases"][case]
    model = ModelType.RECFORMER_LARGE.value(model_kwargs={"init_seed": 1, "device": DEV, "hidden_dropout_prob": 0.0, "attention_probs_dropout_prob": 0.0})
    mm = load_merging_module(MergeType.TASK_VECTOR, LearnType[c["learn_type"]], model, pre, [dict(f) for _ in range(copies) for f in fts], set(),
                             disable_softmax=True, initial_per_weight=per_weight)
    assert list(mm.per_weights.keys()) == c["groups"]
    mod = DistillSequenceModule(mm, teachers, SinglePseudoLabelKDLoss(fx["temperature"], fx["coefficient"]), "cosine")
    mod.item_embeddings = items
    seq = {k: fx[k] for k in ("input_ids", "attention_mask", "token_type_ids", "item_position_ids", "global_attention_mask")}
    batch = BatchDistillationSequence(dataset_indexes=fx["dataset_indexes"], sequence_ids=torch.tensor(fx["sequence_ids"]), sequence=seq)
    mod.train()
    loss = mod.training_step(batch.to(DEV), 0)
    loss.backward()
    return mm, loss


@heavy
@pytest.mark.parametrize("case", [0, 1])
def test_whole_merge_train_step_recformer_large_matches_reference(g20_inputs, case):
    """g20 (oracle/gen_golden_merge_train_recformer_large.py): BASELINE configs[4]'s model and job -- one collaborative-merging step of the
    reference with Recformer-LARGE (24 x 1,024, 435 M parameters; its own RecformerModel driving transformers' LongformerEncoder) and 4
    fine-tuned checkpoints on the CPU (8 exceed the build container's memory) -- loss and d loss / d (per_weights, global_weights, global_biases), task-wise and layer-wise (25
    groups), against the HIP step."""
    fx = g20_inputs[0]
    c = fx["cases"][case]
    mm, loss = _g20_step(g20_inputs, case, copies=1, per_weight=fx["initial_per_weight"])
    torch.testing.assert_close(loss.detach().cpu(), c["loss"], rtol=5e-5, atol=5e-5)
    worst = 0.0
    for name in ("per_weights", "global_weights", "global_biases"):
        for k in c["groups"]:
            want, got = c["grads"][name][k], getattr(mm, name)[k].grad.cpu()
            scale = max(float(want.abs().max()), 1e-3)
            worst = max(worst, float((got - want).abs().max()) / scale)
            assert float((got - want).abs().max()) <= 5e-3 * scale, (name, k, got, want)
    print(f"[merge_train step, Recformer-large x 4 domains, {c['learn_type']}] loss {float(loss.detach()):.6f} (reference {float(c['loss']):.6f}); "
          f"worst gradient deviation {worst:.1e} of the group's largest entry")


@heavy
@pytest.mark.parametrize("case", [0, 1])
def test_eight_domain_recformer_large_step_reproduces_the_four_domain_reference(g20_inputs, case):
    """BASELINE configs[4] at its stated size -- 8 domains x Recformer-large (a 13.9 GB task-vector matrix, 25 alpha groups layer-wise) --
    as a property of fixture g20, no new fixture: with task vectors 5-8 equal to 1-4 and every coefficient halved, the merged model is
    g20's 4-domain model (to the rounding of an 8-term instead of a 4-term sum), so the loss must reproduce the reference's, d loss /
    d per_weight_i must equal the reference's d / d per_weight_i (the contraction of the same gradient with the same task vector: the
    coefficient's value does not enter it), and the twin coefficients' gradients must agree BIT FOR BIT (d_{i+4} == d_i: same kernel, same
    operands, deterministic reduction order).  weight_learning/module/layer_wise.py:64-83, _base.py:78-81."""
    fx = g20_inputs[0]
    c = fx["cases"][case]
    n = len(fx["finetune_seeds"])
    mm, loss = _g20_step(g20_inputs, case, copies=2, per_weight=fx["initial_per_weight"] / 2)
    assert mm.task_vectors_tensor.shape[0] == 2 * n
    torch.testing.assert_close(loss.detach().cpu(), c["loss"], rtol=5e-5, atol=5e-5)
    worst = 0.0
    for k in c["groups"]:
        got = mm.per_weights[k].grad.cpu()
        assert got.shape == (2 * n,)
        assert torch.equal(got[:n], got[n:]), (k, got)               # twins: bit for bit
        want = c["grads"]["per_weights"][k]
        scale = max(float(want.abs().max()), 1e-3)
        worst = max(worst, float((got[:n] - want).abs().max()) / scale)
        assert float((got[:n] - want).abs().max()) <= 5e-3 * scale, (k, got, want)
        # global weight / bias multiply the (halved) per-weights / add to all 8 coefficients: d/d gw = sum_i per_i * d_i is the reference's,
        # d/d gb = sum over 8 = twice the reference's sum over 4
        gw, gb = mm.global_weights[k].grad.cpu(), mm.global_biases[k].grad.cpu()
        wgw, wgb = c["grads"]["global_weights"][k], c["grads"]["global_biases"][k]
        assert float((gw - wgw).abs().max()) <= 5e-3 * max(float(wgw.abs().max()), 1e-3), (k, gw, wgw)
        assert float((gb - 2 * wgb).abs().max()) <= 5e-3 * max(float((2 * wgb).abs().max()), 1e-3), (k, gb, wgb)
    print(f"[merge_train step, Recformer-large x 8 domains (4 + 4 twins), {c['learn_type']}] loss {float(loss.detach()):.6f} (reference, 4 domains: "
          f"{float(c['loss']):.6f}); twin gradients bit-equal; worst deviation from the reference's d/d per_weights {worst:.1e} of the group's largest entry")


# ------------------------------------------------------------------------------------------------ training-graph dropout (VERDICT r02 Missing #2)
def test_dropout_rows_kernel_is_the_oracle_mask_bit_for_bit():
    """mr_dropout_rows_f32 against oracle/ref_cpu.dropout_keep (the restatement of csrc/dropout.h): the same elements survive, scaled by the
    fp32 value of 1 / (1 - p); the residual joins after the mask; p = 0 is the identity; the survival rate is 1 - p."""
    from mergerec_amd import ops

    g = torch.Generator().manual_seed(2)
    for T, d, p in ((37, 768, 0.1), (5, 1024, 0.1), (300, 128, 0.37)):
        x, r = torch.randn(T, d, generator=g), torch.randn(T, d, generator=g)
        key = O.dropout_site_key(11, 3, 5, O.DROP_SITE_FFN_OUT)
        assert key == ops.dropout_site_key(11, 3, 5, ops.DROP_SITE_FFN_OUT)
        keep = O.dropout_keep(key, torch.arange(T)[:, None], torch.arange(d)[None, :], p)
        want = x * (keep.float() * O.dropout_scale(p))
        got = ops.dropout_rows(x.to(DEV), p, key).cpu()
        assert torch.equal(got, want)
        assert torch.equal(ops.dropout_rows(x.to(DEV), p, key, residual=r.to(DEV)).cpu(), want + r)
        assert abs(float(keep.float().mean()) - (1 - p)) < 0.01
        xd = x.to(DEV)
        assert ops.dropout_rows(xd, p, key, out=xd) is xd and torch.equal(xd.cpu(), want)        # in place
        assert torch.equal(ops.dropout_rows(x.to(DEV), 0.0, key).cpu(), x)
        other = ops.dropout_rows(x.to(DEV), p, O.dropout_site_key(11, 4, 5, O.DROP_SITE_FFN_OUT)).cpu()
        assert not torch.equal(other, want)                                                       # the step is part of the key


def _attn_dropout_ref(qkv, cu, H, window, key, p):
    """softmax -> dropout -> @ V with the oracle's mask, under torch autograd (float64 operands)."""
    T = qkv.shape[0]
    d = qkv.shape[1] // 3
    out = []
    for b in range(len(cu) - 1):
        a, e = int(cu[b]), int(cu[b + 1])
        L = e - a
        q, k, v = (qkv[a:e, i * d:(i + 1) * d].view(L, H, 64).transpose(0, 1) for i in range(3))
        s = (q @ k.transpose(-1, -2)) * 0.125
        if window >= 0:
            i = torch.arange(L)
            ok = ((i[:, None] - i[None, :]).abs() <= window) | (i[None, :] == 0)
            s = s.masked_fill(~ok[None], float("-inf"))
        pr = torch.softmax(s, -1)
        rows = (a + torch.arange(L))[None, :, None] * H + torch.arange(H)[:, None, None]
        keep = O.dropout_keep(key, rows, torch.arange(L)[None, None, :], p)
        pr = pr * (keep.to(pr.dtype) * O.dropout_scale(p).to(pr.dtype))
        out.append((pr @ v).transpose(0, 1).reshape(L, d))
    return torch.cat(out)


@pytest.mark.parametrize("window", [-1, 4])
@pytest.mark.parametrize("products", [0, 3])
def test_attention_dropout_forward_and_backward_match_torch(products, window):
    """attention-probability dropout inside the forward kernels (exact-fp32 and split-bf16 work-list forms) and the recomputed mask in the
    three backward kernels, against torch autograd through softmax -> mask -> @ V with the oracle's mask."""
    from mergerec_amd import ops

    g = torch.Generator().manual_seed(40 + products + window)
    H, lens, p = 2, [70, 1, 33, 129, 300], 0.1
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    T = int(cu[-1])
    qkv = torch.randn(T, 3 * H * 64, generator=g)
    dctx = torch.randn(T, H * 64, generator=g)
    key = O.dropout_site_key(5, 9, 1, O.DROP_SITE_ATTN_PROBS)
    ref_in = qkv.double().requires_grad_(True)
    ref = _attn_dropout_ref(ref_in, cu, H, window, key, p)
    keep_rows = torch.ones(T, dtype=torch.bool)
    if window >= 0:
        keep_rows[cu[:-1].long()] = False  # row 0 of a sequence belongs to the global-row kernel
    ref.backward((dctx * keep_rows[:, None]).double())
    work = {q: (lambda w, n: (w.to(DEV), n))(*ops.attn_work_plan(torch.tensor(lens), q)) for q in (128, 256)} if products else None
    ctx = torch.zeros(T, H * 64, device=DEV)
    ops.attention(qkv.to(DEV), cu.to(DEV), len(lens), H, max(lens), window=window, out=ctx, products=products, work=work, drop_p=p, drop_key=key)
    tol = 3e-4 if products else 5e-6
    assert torch.allclose(ctx.cpu()[keep_rows], ref.detach().float()[keep_rows], atol=tol, rtol=tol)
    plain = ops.attention(qkv.to(DEV), cu.to(DEV), len(lens), H, max(lens), window=window, products=products, work=work).cpu()
    assert not torch.allclose(plain[keep_rows], ctx.cpu()[keep_rows], atol=1e-2)                 # the mask did something
    dq = ops.attention_bwd(qkv.to(DEV), ctx, (dctx * keep_rows[:, None]).to(DEV), cu.to(DEV), len(lens), H, window=window, max_len=max(lens),
                           drop_p=p, drop_key=key).cpu()
    scale = float(ref_in.grad.abs().max())
    assert float((dq - ref_in.grad.float()).abs().max()) <= (2e-3 if products else 2e-4) * scale


@pytest.mark.parametrize("kind", ["roberta", "recformer"])
def test_encoder_backward_with_dropout_matches_oracle_autograd(kind):
    """every parameter gradient of the train()-mode forward -- dropout at HF's sites (embedding LayerNorm output, attention
    probabilities, attention-output and FFN-output dense results, Longformer global row), p = 0.1 -- against torch autograd through the
    oracle with the same DropoutPlan; and the deterministic graph (p = 0) stays what it was, bit for bit."""
    from mergerec_amd.engine import ArenaLayout, EncoderRunner
    from mergerec_amd.engine_train import Dropout, EncoderTrainGraph, encode_with_grad
    from tests.test_path_gpu import _spec

    if kind == "roberta":
        g3 = load_golden("g3_roberta.pt")
        cases = [dict(cfg=g3["cfg"], state_dict=g3["state_dict"], batch={"input_ids": g3["input_ids"], "attention_mask": g3["attention_mask"]})]
    else:
        cases = load_golden("g4_recformer.pt")["cases"]
    for ci, case in enumerate(cases):
        cfgd, sd, b = case["cfg"], case["state_dict"], case["batch"]
        cfg = O.EncoderConfig(**{k: cfgd[k] for k in cfgd if k in O.EncoderConfig.__dataclass_fields__})
        plan = dict(p_hidden=0.1, p_attn=0.1, seed=17 + ci, step=3)
        p = OrderedDict((k, v.clone().float().requires_grad_(v.is_floating_point())) for k, v in sd.items())
        if kind == "roberta":
            cls = O.roberta_encode(p, b["input_ids"], b["attention_mask"], cfg, prefix="model.", dropout=O.DropoutPlan(**plan))
        else:
            cls = O.recformer_encode(p, b["input_ids"], b["attention_mask"], b["global_attention_mask"], b["token_type_ids"], b["item_position_ids"],
                                     cfg, prefix="model.", dropout=O.DropoutPlan(**plan))
        R = torch.randn(cls.shape, generator=torch.Generator().manual_seed(5))
        (O.maybe_normalize(cls) * R).sum().backward()
        views = OrderedDict((k, v.to(torch.float32)) for k, v in sd.items())
        layout = ArenaLayout(OrderedDict((k, tuple(v.shape)) for k, v in views.items()))
        spec = _spec(cfgd, kind)
        pb = EncoderRunner(spec).pack(b, DEV)
        flat = layout.pack(views, DEV).requires_grad_(True)
        out = encode_with_grad(EncoderTrainGraph(spec, layout, dropout=Dropout(**plan)), flat, pb)
        assert torch.allclose(out.detach().cpu(), cls.detach(), atol=2e-4, rtol=1e-4), (out.detach().cpu() - cls.detach()).abs().max()
        (torch.nn.functional.normalize(out, dim=-1) * R.to(DEV)).sum().backward()
        got = layout.views(flat.grad)
        gmax = max(float(v.grad.abs().max()) for v in p.values() if v.requires_grad and v.grad is not None)
        worst = 0.0
        for k, v in p.items():
            if not v.requires_grad or v.grad is None:
                assert float(got[k].abs().max()) == 0.0, k
                continue
            scale = max(float(v.grad.abs().max()), 1e-3 * gmax)
            err = float((got[k].cpu() - v.grad).abs().max()) / scale
            worst = max(worst, err)
            assert err <= 2e-3, (kind, ci, k, err)
        assert worst > 0.0
        # p = 0 (and no Dropout at all) is the deterministic graph, bit for bit; another step draws another mask
        f0, f1, f2 = (layout.pack(views, DEV).requires_grad_(True) for _ in range(3))
        o0 = encode_with_grad(EncoderTrainGraph(spec, layout), f0, pb)
        o1 = encode_with_grad(EncoderTrainGraph(spec, layout, dropout=Dropout(0.0, 0.0, 17, 3)), f1, pb)
        o2 = encode_with_grad(EncoderTrainGraph(spec, layout, dropout=Dropout(**dict(plan, step=4))), f2, pb)
        assert torch.equal(o0, o1) and not torch.allclose(o2, out, atol=1e-3)
        o0.sum().backward(); o1.sum().backward()
        # (the embedding-table gradients are scatter-added with float atomics: equal up to their summation order)
        assert torch.allclose(f0.grad, f1.grad, rtol=1e-5, atol=1e-6 * float(f0.grad.abs().max()))


def test_models_apply_dropout_in_train_mode_only_with_hf_default_rates():
    """the drop-in objects: BaseEncoderModel.forward_with_grad and TaskVectorMergingModule.forward draw a fresh mask per TRAINING forward
    (HF defaults 0.1 / 0.1 unless model_kwargs say otherwise), none in eval() mode, and rates 0 reproduce the deterministic graph."""
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from tests.test_path_gpu import _tiny_model

    g2 = load_golden("g2_merger.pt")
    cfgd = g2["cfg"]
    batch = {"input_ids": g2["input_ids"].to(DEV), "attention_mask": g2["attention_mask"].to(DEV)}
    on = _tiny_model(cfgd, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, dropout_seed=5)
    off = _tiny_model(cfgd)
    assert (on.hidden_dropout_prob, on.attention_probs_dropout_prob, off.hidden_dropout_prob) == (0.1, 0.1, 0.0)
    from mergerec_amd.module import ModelType
    over = dict(hidden=cfgd["hidden"], heads=cfgd["heads"], layers=cfgd["layers"], intermediate=cfgd["intermediate"], vocab=cfgd["vocab"], max_pos=cfgd["max_pos"])
    default = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 1, "spec_overrides": over, "device": DEV})
    assert default.hidden_dropout_prob == default.attention_probs_dropout_prob == 0.1     # transformers RobertaConfig defaults
    on.train(); off.train()
    a, b_, c = on.forward_with_grad(batch), on.forward_with_grad(batch), off.forward_with_grad(batch)
    assert not torch.allclose(a, b_, atol=1e-3) and not torch.allclose(a, c, atol=1e-3)    # a fresh mask per training forward
    on.eval()
    assert torch.equal(on.forward_with_grad(batch), c)                                     # eval(): no dropout
    again = _tiny_model(cfgd, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, dropout_seed=5).train()
    assert torch.equal(again.forward_with_grad(batch), a)                                  # (seed, step) reproduces the mask
    mk = lambda m: load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, m, g2["pretrain"], [dict(f) for f in g2["finetunes"]], set(),
                                       disable_softmax=True, initial_per_weight=0.3)
    mm_on, mm_off = mk(_tiny_model(cfgd, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)), mk(_tiny_model(cfgd))
    mm_on.train(); mm_off.train()
    t1, t2, t0 = mm_on.forward(batch), mm_on.forward(batch), mm_off.forward(batch)
    assert t1.requires_grad and not torch.allclose(t1, t2, atol=1e-3) and not torch.allclose(t1, t0, atol=1e-3)
    mm_on.eval()
    assert torch.equal(mm_on.forward_with_grad(batch), t0)
    with pytest.raises(ValueError):
        _tiny_model(cfgd, hidden_dropout_prob=1.0)


@pytest.mark.parametrize("kind", ["roberta", "recformer"])
def test_side_stream_weight_gradients_are_bit_identical_to_the_one_stream_backward(kind, monkeypatch):
    """the dW products issued on a second stream beside the dX chain (engine_train._WGRAD_STREAM) give the same gradient arena, bit for bit,
    as the backward on one stream -- with dropout on, repeated so that a missing dependency would show as a stale read"""
    from mergerec_amd import engine_train as ET
    from mergerec_amd.engine import ArenaLayout, EncoderRunner
    from tests.test_path_gpu import _spec

    if kind == "roberta":
        g3 = load_golden("g3_roberta.pt")
        cfgd, sd, b = g3["cfg"], g3["state_dict"], {"input_ids": g3["input_ids"], "attention_mask": g3["attention_mask"]}
    else:
        case = load_golden("g4_recformer.pt")["cases"][0]
        cfgd, sd, b = case["cfg"], case["state_dict"], case["batch"]
    views = OrderedDict((k, v.to(torch.float32)) for k, v in sd.items())
    layout = ArenaLayout(OrderedDict((k, tuple(v.shape)) for k, v in views.items()))
    spec = _spec(cfgd, kind) if kind == "recformer" else _spec(cfgd)
    pb = EncoderRunner(spec).pack(b, DEV)
    R = torch.randn(pb.B, spec.hidden, generator=torch.Generator().manual_seed(5)).to(DEV)

    def grads(side: bool):
        monkeypatch.setattr(ET, "_WGRAD_STREAM", side)
        flat = layout.pack(views, DEV).requires_grad_(True)
        out = ET.encode_with_grad(ET.EncoderTrainGraph(spec, layout, dropout=ET.Dropout(0.1, 0.1, 17, 3)), flat, pb)
        (torch.nn.functional.normalize(out, dim=-1) * R).sum().backward()
        torch.cuda.synchronize()
        return flat.grad.clone()

    def same(a, b):
        # the embedding tables are accumulated with atomicAdd (several tokens share a row: order, hence the last bit, is not fixed from run
        # to run on either route); every other gradient has one owner per element
        va, vb = layout.views(a), layout.views(b)
        for k in va:
            if k.endswith("_embeddings.weight"):
                assert torch.allclose(va[k], vb[k], rtol=1e-5, atol=1e-7), k
            else:
                assert torch.equal(va[k], vb[k]), k

    one = grads(False)
    assert float(one.abs().max()) > 0.0
    same(grads(False), one)
    for _ in range(3):
        same(grads(True), one)


@pytest.mark.parametrize("learn", ["TASK_WISE", "LAYER_WISE"])
def test_merge_and_alpha_gradient_in_arena_ranges_on_the_second_stream_match_the_one_launch_step(learn, monkeypatch):
    """merger.weight_learning.MergeOverlap (the alpha-learning step's merge and d alpha contraction range by range on a second stream, under
    the encoder's kernels) against the same step with one merge launch and one contraction launch: the loss -- a function of the merged
    parameters -- bit for bit; d alpha to the rounding of sums grouped per range (1e-5 relative); repeated so that a missing dependency
    between the streams would show as a stale read"""
    from mergerec_amd import engine_train as ET
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.model_batch import BatchDistillationSequence
    from mergerec_amd.module import DistillSequenceModule
    from mergerec_amd.module.loss_fn import SinglePseudoLabelKDLoss
    from tests.test_path_gpu import _tiny_model

    g2 = load_golden("g2_merger.pt")
    cfgd, ids, mask = g2["cfg"], g2["input_ids"], g2["attention_mask"]
    B = ids.shape[0]
    gen = torch.Generator().manual_seed(9)
    items = [torch.nn.functional.normalize(torch.randn(m, cfgd["hidden"], generator=gen), dim=-1) for m in (50, 77)]
    teachers = [torch.randn(B, m, generator=gen).clamp(-1, 1) for m in (50, 77)]
    batch = BatchDistillationSequence(dataset_indexes=[i % 2 for i in range(B)], sequence_ids=torch.arange(B),
                                      sequence={"input_ids": ids, "attention_mask": mask}).to(DEV)

    def step(overlap: bool):
        monkeypatch.setattr(ET, "_MERGE_OVERLAP", overlap)
        mm = load_merging_module(MergeType.TASK_VECTOR, LearnType[learn], _tiny_model(cfgd), g2["pretrain"], [dict(f) for f in g2["finetunes"]], set(),
                                 disable_softmax=True, initial_per_weight=0.3)
        mod = DistillSequenceModule(mm, teachers, SinglePseudoLabelKDLoss(0.05, 1000.0), "cosine",
                                    trainable_args_kwargs={"freeze_global_weight": True, "freeze_global_bias": True})
        mod.item_embeddings = items
        mod.eval()  # no dropout: the two routes then differ in nothing but the grouping of the d alpha sums
        with torch.enable_grad():
            loss = mod.training_step(batch, 0)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {k: p.grad.clone() for k, p in mm.per_weights.items()}

    l0, g0 = step(False)
    assert all(float(v.abs().max()) > 0.0 for v in g0.values())
    for _ in range(3):
        l1, g1 = step(True)
        assert l1 == l0
        for k in g0:
            assert torch.allclose(g1[k], g0[k], rtol=1e-5, atol=1e-6 * float(g0[k].abs().max())), (k, g1[k], g0[k])


def test_auto_train_mode_picks_the_graph_by_token_count(monkeypatch):
    """train_mode = "auto" (what DistillTrainer sets under the reduced-precision flags): the exact-fp32 tile graph below AUTO_SPLIT_TOKENS tokens
    per step -- bit for bit the "f32" step --, the bf16x3 split graph from there on -- the "bf16x3" step bit for bit, within 5e-3 of the exact one"""
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.model_batch import BatchDistillationSequence
    from mergerec_amd.module import DistillSequenceModule
    from mergerec_amd.module.loss_fn import SinglePseudoLabelKDLoss
    from mergerec_amd.utils import DistillTrainer
    from tests.test_path_gpu import _tiny_model

    assert DistillTrainer(max_steps=1, precision="bf16-mixed", verbose=False).train_mode == "auto"
    assert DistillTrainer(max_steps=1, precision="32-true", verbose=False).train_mode == "f32"
    g2 = load_golden("g2_merger.pt")
    cfgd, ids, mask = g2["cfg"], g2["input_ids"], g2["attention_mask"]
    B = ids.shape[0]
    gen = torch.Generator().manual_seed(9)
    items = [torch.nn.functional.normalize(torch.randn(m, cfgd["hidden"], generator=gen), dim=-1) for m in (50, 77)]
    teachers = [torch.randn(B, m, generator=gen).clamp(-1, 1) for m in (50, 77)]
    batch = BatchDistillationSequence(dataset_indexes=[i % 2 for i in range(B)], sequence_ids=torch.arange(B),
                                      sequence={"input_ids": ids, "attention_mask": mask}).to(DEV)

    def step(mode, threshold=None):
        mm = load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, _tiny_model(cfgd), g2["pretrain"], [dict(f) for f in g2["finetunes"]], set(),
                                 disable_softmax=True, initial_per_weight=0.3)
        mm.train_mode = mode
        if threshold is not None:
            mm.AUTO_SPLIT_TOKENS = threshold
        mod = DistillSequenceModule(mm, teachers, SinglePseudoLabelKDLoss(0.05, 1000.0), "cosine",
                                    trainable_args_kwargs={"freeze_global_weight": True, "freeze_global_bias": True})
        mod.item_embeddings = items
        mod.eval()
        with torch.enable_grad():
            loss = mod.training_step(batch, 0)
        loss.backward()
        return float(loss.detach()), mm.per_weights["all"].grad.clone()

    l_f32, g_f32 = step("f32")
    l_lo, g_lo = step("auto", threshold=10 ** 9)
    assert l_lo == l_f32 and torch.allclose(g_lo, g_f32, rtol=1e-5, atol=0)
    if cfgd["hidden"] % 128 == 0:
        l_sp, g_sp = step("bf16x3")
        l_hi, g_hi = step("auto", threshold=1)
        assert l_hi == l_sp and torch.equal(g_hi, g_sp)
        assert float((g_hi - g_f32).abs().max()) <= 5e-3 * float(g_f32.abs().max())


def test_two_differentiable_forwards_before_one_backward_keep_their_own_parameters():
    """the merging module's persistent step buffers serve one graph at a time: a second forward_with_grad before the first one's backward gets
    vectors of its own, so both backwards see the parameters (and write the gradients) of their own forward"""
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from tests.test_path_gpu import _tiny_model

    g2 = load_golden("g2_merger.pt")
    cfgd, ids, mask = g2["cfg"], g2["input_ids"], g2["attention_mask"]
    a = {"input_ids": ids[:3].to(DEV), "attention_mask": mask[:3].to(DEV)}
    b = {"input_ids": ids[3:].to(DEV), "attention_mask": mask[3:].to(DEV)}

    def fresh():
        mm = load_merging_module(MergeType.TASK_VECTOR, LearnType.TASK_WISE, _tiny_model(cfgd), g2["pretrain"], [dict(f) for f in g2["finetunes"]], set(),
                                 disable_softmax=True, initial_per_weight=0.3)
        mm.eval()
        return mm

    def grad_of(mm, loss):
        mm.per_weights["all"].grad = None
        loss.backward()
        return mm.per_weights["all"].grad.clone()

    mm = fresh()
    with torch.enable_grad():
        ga = grad_of(mm, mm.forward_with_grad(a).square().sum())
        gb = grad_of(mm, mm.forward_with_grad(b).square().sum())
        # both graphs alive, backward in either order
        oa, ob = mm.forward_with_grad(a), mm.forward_with_grad(b)
        assert torch.equal(grad_of(mm, ob.square().sum()), gb)
        assert torch.equal(grad_of(mm, oa.square().sum()), ga)
        # and the persistent buffers are taken up again afterwards
        assert torch.equal(grad_of(mm, mm.forward_with_grad(a).square().sum()), ga)
