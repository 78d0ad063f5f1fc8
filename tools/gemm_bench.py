"""Micro-benchmark of mr_gemm_nt_bias_act_f32 at the encoder's shapes (HIP events, interleaved rounds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergerec_amd import ops

dev = "cuda:0"
M = int(os.environ.get("GB_M", 65536))
shapes = [("qkv", 768, 768, 3), ("out", 768, 768, 1), ("ffn1", 3072, 768, 1), ("ffn2", 768, 3072, 1)]
rounds = int(os.environ.get("GB_ROUNDS", 5))
g = torch.Generator(device=dev).manual_seed(0)
bufs = {}
for name, n, k, nseg in shapes:
    A = torch.randn(M, k, device=dev, generator=g)
    Ws = [torch.randn(n, k, device=dev, generator=g) * 0.02 for _ in range(nseg)]
    bs = [torch.randn(n, device=dev, generator=g) for _ in range(nseg)]
    out = torch.empty(M, n * nseg, device=dev)
    bufs[name] = (A, Ws, bs, out)
for name, n, k, nseg in shapes:  # warm
    A, Ws, bs, out = bufs[name]; ops.gemm_nt(A, Ws, bs, out=out)
torch.cuda.synchronize()
res = {s[0]: [] for s in shapes}
for r in range(rounds):
    for name, n, k, nseg in shapes:
        A, Ws, bs, out = bufs[name]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gemm_nt(A, Ws, bs, out=out); e1.record(); torch.cuda.synchronize()
        res[name].append(e0.elapsed_time(e1))
for name, n, k, nseg in shapes:
    ms = sorted(res[name])[len(res[name]) // 2]
    print(f"{name:5s} M={M} N={n*nseg} K={k}: {ms:.3f} ms  {2.0*M*n*nseg*k/ms/1e9:.1f} TFLOP/s (min {2.0*M*n*nseg*k/min(res[name])/1e9:.1f})")
