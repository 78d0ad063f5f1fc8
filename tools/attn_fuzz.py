"""Randomised differential test of the attention kernels (fp32 / bf16x6 / bf16x3, full and Longformer-windowed, backward kernels)
on ragged batches against a float64 torch reference, through BOTH launch forms: the work list the engine builds while packing (the
product's route) and the (blocks, H, B) box grid -- which must also agree with each other bit for bit.  Exit code 1 on any mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergerec_amd import ops

dev = "cuda:0"
g = torch.Generator().manual_seed(int(os.environ.get("FZ_SEED", 0)))
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
for it in range(int(os.environ.get("FZ_N", 24))):
    H = [1, 2, 12][ri(0, 2)]
    B = ri(1, 9)
    lens = [[1, 2, 31, 32, 33, 127, 128, 129, 512, ri(1, 520)][ri(0, 9)] for _ in range(B)]
    window = -1 if ri(0, 1) else [32, 5][ri(0, 1)]
    T = sum(lens)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    qkv = torch.randn(T, 3 * H * 64, generator=g)
    q = qkv.double().clone().requires_grad_(True)
    dctx = torch.randn(T, H * 64, generator=g)
    outs = []
    for b in range(B):
        s, e = int(cu[b]), int(cu[b + 1])
        L = e - s
        Q, K, V = (q[s:e, i * H * 64:(i + 1) * H * 64].view(L, H, 64).transpose(0, 1) for i in range(3))
        S = Q @ K.transpose(1, 2) * 0.125
        if window >= 0:
            i_ = torch.arange(L)[:, None]; j_ = torch.arange(L)[None, :]
            ok = (j_ == 0) | ((i_ - j_).abs() <= window)
            S = S.masked_fill(~ok[None], float("-inf"))
        outs.append((torch.softmax(S, dim=-1) @ V).transpose(0, 1).reshape(L, H * 64))
    ref = torch.cat(outs)
    rows = torch.ones(T, dtype=torch.bool)
    if window >= 0:
        rows[cu[:-1].long()] = False  # row 0 of every sequence belongs to the global-row kernel
    work = {q: (lambda w, n: (w.to(dev), n))(*ops.attn_work_plan(torch.tensor(lens), q)) for q in (128, 256)}
    for products, tol in ((0, 3e-6), (6, 3e-6), (3, 2e-4)):
        box = ops.attention(qkv.to(dev), cu.to(dev), B, H, max(lens), window=window, products=products)
        lst = ops.attention(qkv.to(dev), cu.to(dev), B, H, max(lens), window=window, products=products, work=work)
        if not torch.equal(box[rows.to(dev)], lst[rows.to(dev)]):
            bad += 1
            print(f"MISMATCH work list vs box grid it={it} products={products} H={H} lens={lens} window={window}")
        got = lst.cpu().double()
        err = float((got[rows] - ref.detach()[rows]).abs().max()) if rows.any() else 0.0
        if not err <= tol:
            bad += 1
            print(f"MISMATCH fwd it={it} products={products} H={H} lens={lens} window={window}: {err:.3e}")
    # backward (fp32 kernels): gradient of sum(ctx * dctx) over the rows the kernel owns
    dd = dctx.clone(); dd[~rows] = 0
    (ref * dd.double()).sum().backward()
    ctx_dev = ops.attention(qkv.to(dev), cu.to(dev), B, H, max(lens), window=window, products=0)
    gq_box = ops.attention_bwd(qkv.to(dev), ctx_dev, dd.to(dev), cu.to(dev), B, H, window=window, max_len=max(lens))
    gq_dev = ops.attention_bwd(qkv.to(dev), ctx_dev, dd.to(dev), cu.to(dev), B, H, window=window, max_len=max(lens), work=work)
    if not torch.equal(gq_box, gq_dev):
        bad += 1
        print(f"MISMATCH bwd work list vs box grid it={it} H={H} lens={lens} window={window}")
    gq = gq_dev.cpu().double()
    err = float((gq - q.grad).abs().max()) / (float(q.grad.abs().max()) + 1e-9)
    if not err <= 2e-5:
        bad += 1
        print(f"MISMATCH bwd it={it} H={H} lens={lens} window={window}: {err:.3e}")
print("attention fuzz done:", "OK" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
