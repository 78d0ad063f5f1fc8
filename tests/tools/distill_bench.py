"""Micro-benchmark of the fused distillation row loss (mr_distill_loss_rows_f32) against the CPU oracle.
cfg5 shape: 16 rows per step over a catalog of M items (SINGLE_PSEUDO_LABEL_KD, T = 0.05, coefficient = 1000); a large-batch
shape shows the HBM-bound regime.  Algorithmic bytes per row: z and t read once, dz written once = 12 M bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mergerec_amd import ops
from oracle import ref_cpu as O

dev = "cuda:0"
spec = dict(label_src=1, w_ce=1.0, w_kd=1000.0, temperature=0.05)
g = torch.Generator().manual_seed(0)
for rows, M in [(16, 22855), (16, 3686), (4096, 22855)]:
    z = (torch.randn(rows, M, generator=g) * 0.3).clamp(-1, 1)
    t = (z * 0.7 + torch.randn(rows, M, generator=g) * 0.15).clamp(-1, 1)
    zd, td = z.to(dev), t.to(dev)
    dz = torch.empty_like(zd)
    for _ in range(3):
        ops.distill_loss_rows(zd, td, dz=dz, grad_scale=1.0 / rows, **spec)
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lr, _ = ops.distill_loss_rows(zd, td, dz=dz, grad_scale=1.0 / rows, **spec); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    n_cpu = min(rows, 64)
    torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
    zc = z[:n_cpu].clone().requires_grad_(True)
    t0 = time.perf_counter()
    for i in range(n_cpu):  # the reference's per-sample loop (module/distiller/sequence/module.py:62-72)
        l = O.distill_loss("SINGLE_PSEUDO_LABEL_KD", zc[i:i + 1], t[i:i + 1], 0.05, 1000.0)
        l.backward()
    cpu_ms = (time.perf_counter() - t0) * 1e3 / n_cpu * rows
    ref = torch.stack([O.distill_loss("SINGLE_PSEUDO_LABEL_KD", z[i:i + 1], t[i:i + 1], 0.05, 1000.0) for i in range(n_cpu)])
    err = ((lr[:n_cpu].cpu() - ref).abs() / ref.abs().clamp_min(1e-6)).max().item()
    print(f"rows={rows} M={M}: {ms*1e3:.1f} us  {12.0*rows*M/ms/1e6:.1f} GB/s algorithmic  |  CPU oracle loop (fwd+bwd, {torch.get_num_threads()} threads) {cpu_ms:.2f} ms  -> x{cpu_ms/ms:.0f}  | max rel loss err {err:.2e}")
