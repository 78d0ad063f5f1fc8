"""alpha-gradient kernel (mr_merge_bwd_alpha_f32) at BLaIR-base / Recformer-large size, N = 8: device time and the streams' rate, for the
single-pass kernel and (mr_merge_bwd_generic(1)) the per-vector loop.   PYTHONPATH=. python tools/merge_bwd_bench.py"""
import os
import torch
from mergerec_amd import _lib, ops

dev = torch.device("cuda:0")
for name, P, S in (("BLaIR-base task-wise", 124645632, 1), ("BLaIR-base layer-wise (13 groups)", 124645632, 13), ("Recformer-large task-wise", 433610752, 1)):
    N = 8
    tv, g = torch.randn(N, P, device=dev), torch.randn(P, device=dev)
    seg = None
    if S > 1:
        cut = torch.linspace(0, P // 64, S + 1).long() * 64
        cut[-1] = P
        seg = cut.to(dev)
    for mode in ("single pass", "per-vector loop"):
        _lib.load().mr_merge_bwd_generic(1 if mode == "per-vector loop" else 0)
        for _ in range(3):
            out = ops.merge_bwd_alpha(tv, g, seg)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            out = ops.merge_bwd_alpha(tv, g, seg)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 20
        print(f"{name:36s} {mode:16s} {ms:6.3f} ms  {(N + 1) * P * 4 / ms / 1e9:6.2f} TB/s  checksum {float(out.double().sum()):.6e}", flush=True)
    _lib.load().mr_merge_bwd_generic(0)
    del tv, g
