"""Randomised differential test of the three GEMM kernels (edge tiles, strided outputs, bias / GELU / residual, 1-3 segments)
against a float64 torch reference.  Exit code 1 on any mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergerec_amd import ops

dev = "cuda:0"
g = torch.Generator().manual_seed(int(os.environ.get("FZ_SEED", 0)))
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
for it in range(int(os.environ.get("FZ_N", 60))):
    nseg = ri(1, 3)
    K = 16 * ri(1, 40)
    seg_n = 128 * ri(1, 5) if nseg > 1 else (ri(1, 700) if ri(0, 1) else 256 * ri(1, 4))
    M = ri(1, 1500) if ri(0, 3) else ri(1, 40)
    lda = K + 4 * ri(0, 3)
    ldc = nseg * seg_n + ri(0, 9)
    act = ops.ACT_GELU if ri(0, 2) == 0 else ops.ACT_NONE
    use_r, use_b = ri(0, 1), ri(0, 1)
    A = torch.randn(M, lda, generator=g)[:, :K]
    Ws = [torch.randn(seg_n, K, generator=g) * 0.1 for _ in range(nseg)]
    bs = [torch.randn(seg_n, generator=g) if use_b else None for _ in range(nseg)]
    R = torch.randn(M, ldc + 3, generator=g)[:, : nseg * seg_n] if use_r else None
    ref = torch.cat([A.double() @ w.double().T + (b.double() if b is not None else 0) for w, b in zip(Ws, bs)], dim=1)
    if act == ops.ACT_GELU:
        ref = torch.nn.functional.gelu(ref)
    if R is not None:
        ref = ref + R.double()
    Ad = torch.empty(M, lda, device=dev); Ad[:, :K] = A.to(dev); Ad = Ad[:, :K]
    Rd = None
    if R is not None:
        Rd = torch.empty(M, ldc + 3, device=dev); Rd[:, : nseg * seg_n] = R.to(dev); Rd = Rd[:, : nseg * seg_n]
    flat = torch.cat([w.reshape(-1) for w in Ws]).to(dev)
    Wd = [flat[i * seg_n * K:(i + 1) * seg_n * K].view(seg_n, K) for i in range(nseg)]
    bd = [b.to(dev) if b is not None else None for b in bs]
    scale = float(ref.abs().max()) + 1e-6
    for mode, tol in (("f32", 2e-5), ("bf16x6", 2e-5), ("bf16x3", 3e-4)):
        out = torch.full((M, ldc), float("nan"), device=dev)
        o = out[:, : nseg * seg_n]
        if mode == "f32":
            ops.gemm_nt(Ad, Wd, bd, act, Rd, out=o)
        else:
            pieces = ops.split_weights_kblock(flat, ops.KBlockTable([(i * seg_n * K, seg_n, K) for i in range(nseg)], dev))
            ops.gemm_nt_split(Ad, pieces, [i * seg_n * K for i in range(nseg)], seg_n, K, bd, act, Rd, out=o, products=6 if mode == "bf16x6" else 3)
        err = float((o.cpu().double() - ref).abs().max()) / scale
        pad_ok = bool(torch.isnan(out[:, nseg * seg_n:]).all())
        if not (err <= tol) or not pad_ok:
            bad += 1
            print(f"MISMATCH it={it} mode={mode} M={M} seg_n={seg_n} nseg={nseg} K={K} lda={lda} ldc={ldc} act={act} R={use_r} b={use_b}: rel err {err:.3e} pad_ok={pad_ok}")
    if M >= 1 and nseg == 1:  # split-K entry point of the training graph
        out = torch.full((M, ldc), float("nan"), device=dev)
        ops.gemm_nt_train(Ad, Wd[0], bd[0], Rd if act == ops.ACT_NONE else None, out=out[:, :seg_n])
        ref2 = A.double() @ Ws[0].double().T + (bs[0].double() if bs[0] is not None else 0) + (R.double() if (R is not None and act == ops.ACT_NONE) else 0)
        err = float((out[:, :seg_n].cpu().double() - ref2).abs().max()) / (float(ref2.abs().max()) + 1e-6)
        if not err <= 2e-5 or not bool(torch.isnan(out[:, seg_n:]).all()):
            bad += 1
            print(f"MISMATCH splitk it={it} M={M} N={seg_n} K={K}: {err:.3e}")
print("fuzz done:", "OK" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
