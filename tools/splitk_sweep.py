"""Split-K sweep of the exact-fp32 training GEMM (ops.gemm_nt_train) at the alpha-learning step's shapes: device time per call
(product + partial-sum launch) for each split count beside the heuristic's choice.   PYTHONPATH=. python tools/splitk_sweep.py [tokens]"""
import sys
import torch
from mergerec_amd import ops

T = int(sys.argv[1]) if len(sys.argv) > 1 else 602
Tp = (T + 15) // 16 * 16
dev = torch.device("cuda:0")
import os

H = int(os.environ.get("SW_HIDDEN", 768))  # SW_HIDDEN=1024: the large models' shapes
shapes = [("fwd q/k/v/out", T, H, H), ("fwd up", T, 4 * H, H), ("fwd down", T, H, 4 * H), ("dgrad qkv", T, H, 3 * H),
          ("dgrad up", T, H, 4 * H), ("dgrad down", T, 4 * H, H), ("wgrad d x d", H, H, Tp), ("wgrad up", 4 * H, H, Tp),
          ("wgrad down", H, 4 * H, Tp)]


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, M, N, K in ([] if (len(sys.argv) > 2 and sys.argv[2] == "bf16x3") else shapes):
    A, W = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    auto = ops.splitk_plan(M, N, K, False)
    res = {}
    for s in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16, 24, 32):
        if s > K // 16:
            continue
        res[s] = timed(lambda: ops.gemm_nt_train(A, W, out=out, splits=s))
    best = min(res, key=res.get)
    print(f"{name:16s} M={M:5d} N={N:5d} K={K:5d} tiles={tiles:3d} auto={auto:2d} ({timed(lambda: ops.gemm_nt_train(A, W, out=out)):6.1f} us)  best={best:2d} "
          f"({res[best]:6.1f} us)  " + " ".join(f"{s}:{t:.1f}" for s, t in res.items()), flush=True)


def sweep_bf16x3(tokens):
    """the fine-tuning weight gradients: dW (N_out, N_in) = dY^T (N_out, T_pad) x pieces of X^T (N_in, T_pad), bf16x3 split-K kernel"""
    for name, M, N in (("wgrad 768x768", 768, 768), ("wgrad up", 3072, 768), ("wgrad down", 768, 3072)):
        x = torch.randn(tokens, N, device=dev)
        pieces, tp = ops.split_tokens_kblock(x, pad=32)
        A = torch.randn(M, tp, device=dev)
        out = torch.empty(M, N, device=dev)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        auto = max(1, min(tp // 64, 64, 1024 // max(tiles, 1)))
        res = {}
        for s in (2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32, 40, 48, 56, 64):
            res[s] = timed(lambda: ops.gemm_nt_split_k(A, pieces, 0, N, tp, out=out, splits=s), n=10)
        best = min(res, key=res.get)
        t_auto = timed(lambda: ops.gemm_nt_split_k(A, pieces, 0, N, tp, out=out), n=10)
        fl = 2.0 * M * N * tp / 1e6
        print(f"{name:14s} M={M:5d} N={N:5d} K={tp:6d} tiles={tiles:3d} auto={auto:2d} ({t_auto:6.1f} us, {fl / t_auto:5.0f} TFLOP/s)  best={best:2d} "
              f"({res[best]:6.1f} us, {fl / res[best]:5.0f} TFLOP/s)  " + " ".join(f"{s}:{t:.0f}" for s, t in res.items()), flush=True)


if len(sys.argv) > 2 and sys.argv[2] == "bf16x3":
    sweep_bf16x3(T)
