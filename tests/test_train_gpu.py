"""Encoder backward (csrc/backward.hip + engine_train.py) against torch autograd through the CPU oracle restatement of the
reference's RoBERTa forward (oracle/ref_cpu.py, itself pinned by the transformers golden g3_roberta.pt)."""
from collections import OrderedDict

import pytest
import torch

from oracle import ref_cpu as O
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_backward_kernels_match_torch():
    from mergerec_amd import ops

    g = torch.Generator().manual_seed(3)
    x = torch.randn(37, 100, generator=g)
    t = ops.transpose_pad(x.to(DEV)).cpu()
    assert t.shape == (100, 48) and torch.equal(t[:, :37], x.T) and torch.count_nonzero(t[:, 37:]) == 0
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
