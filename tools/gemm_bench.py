"""Micro-benchmark of the encoder GEMMs at BLaIR-base shapes (HIP events, interleaved rounds).
GB_MODE = f32 | bf16x6 | bf16x3 | f16x3 (comma-separated; default f32,bf16x6)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergerec_amd import ops

dev = "cuda:0"
M = int(os.environ.get("GB_M", 65536))
shapes = [("qkv", 768, 768, 3), ("out", 768, 768, 1), ("ffn1", 3072, 768, 1), ("ffn2", 768, 3072, 1)]
rounds = int(os.environ.get("GB_ROUNDS", 5))
modes = os.environ.get("GB_MODE", "f32,bf16x6").split(",")
g = torch.Generator(device=dev).manual_seed(0)
bufs = {}
for name, n, k, nseg in shapes:
    A = torch.randn(M, k, device=dev, generator=g)
    W = torch.randn(nseg * n * k, device=dev, generator=g) * 0.02
    Ws = [W[i * n * k:(i + 1) * n * k].view(n, k) for i in range(nseg)]
    bs = [torch.randn(n, device=dev, generator=g) for _ in range(nseg)]
    out = torch.empty(M, n * nseg, device=dev)
    tab = ops.KBlockTable([(i * n * k, n, k) for i in range(nseg)], dev)
    bufs[name] = (A, W, Ws, bs, out, ops.split_weights_kblock(W, tab), ops.split_weights_kblock(W, tab, f16=True))

def run(mode, name, n, k, nseg):
    A, W, Ws, bs, out, pieces, pieces_h = bufs[name]
    if mode == "f32":
        ops.gemm_nt(A, Ws, bs, out=out)
    elif mode == "f16x3":
        ops.gemm_nt_split(A, pieces_h, [i * n * k for i in range(nseg)], n, k, bs, out=out, products=ops.PRODUCTS_F16X3)
    else:
        ops.gemm_nt_split(A, pieces, [i * n * k for i in range(nseg)], n, k, bs, out=out, products=6 if mode == "bf16x6" else 3)

for mode in modes:
    for s in shapes:
        run(mode, *s)
torch.cuda.synchronize()
res = {(m, s[0]): [] for m in modes for s in shapes}
for r in range(rounds):
    for mode in modes:
        for s in shapes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(mode, *s); e1.record(); torch.cuda.synchronize()
            res[(mode, s[0])].append(e0.elapsed_time(e1))
for mode in modes:
    for name, n, k, nseg in shapes:
        t = res[(mode, name)]
        ms = sorted(t)[len(t) // 2]
        print(f"{mode:7s} {name:5s} M={M} N={n*nseg} K={k}: {ms:.3f} ms  {2.0*M*n*nseg*k/ms/1e9:.1f} TFLOP/s (best {2.0*M*n*nseg*k/min(t)/1e9:.1f})")
