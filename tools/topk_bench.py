"""Row select (mr_topk_rows_f32) on cache-resident score blocks: HIP-event time per launch.  MR_TOPK_LDS=1 selects the r02 LDS kernel.
Usage: python tools/topk_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergerec_amd import ops

dev = "cuda:0"
g = torch.Generator().manual_seed(0)
tag = "lds (r02)" if os.environ.get("MR_TOPK_LDS") == "1" else "registers"
for R, C, k in ((256, 22855, 50), (256, 4968, 50), (32, 18357, 50), (1024, 22855, 50), (256, 27932, 50), (256, 22855, 200)):
    if k > 64 and tag != "registers":
        continue
    base = torch.randn(1, 768, generator=g)
    U = torch.nn.functional.normalize(base + 0.7 * torch.randn(R, 768, generator=g), dim=1)
    E = torch.nn.functional.normalize(base + 0.7 * torch.randn(C, 768, generator=g), dim=1)
    ld = (C + 3) // 4 * 4
    s = torch.zeros(R, ld)
    s[:, :C] = U @ E.T
    s = s.to(dev)[:, :C]
    labels = torch.randint(0, C, (R,), generator=g).to(dev)
    for _ in range(3):
        ops.topk_rows(s, k, labels, 20.0)
    torch.cuda.synchronize()
    reps = 30
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.topk_rows(s, k, labels, 20.0)
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag:10s} rows {R:5d} x cols {C:6d} k={k:4d}: {e0.elapsed_time(e1) / reps * 1e3:8.1f} us", flush=True)
