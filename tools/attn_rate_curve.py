"""Rate of the split-bf16 attention kernel against sequence length: uniform batches of ~64 k tokens at several L, and the bench's ragged user
batch -- shows where the ragged batch loses against uniform L = 512 (HIP events, median of AB_ROUNDS launches).
Usage: python tools/attn_rate_curve.py [products]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergerec_amd import ops
from mergerec_amd.synthetic import blair_sequence_lengths

dev, H = "cuda:0", 12
prod = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def run(lens, label):
    B = lens.numel()
    cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0)
    T = int(cu[-1])
    qkv = torch.randn(T, 3 * H * 64, device=dev)
    cu_d, out = cu.to(dev), torch.empty(T, H * 64, device=dev)
    order = torch.argsort(lens, descending=True, stable=True).to(torch.int32).to(dev)
    flops = 4.0 * 768 * float((lens.double() ** 2).sum())
    work = None
    if os.environ.get("AB_WORKLIST", "1") == "1":
        work = {q: (lambda w, n: (w.to(dev), n))(*ops.attn_work_plan(lens, q)) for q in (128, 256)}
    f = lambda: ops.attention(qkv, cu_d, B, H, int(lens.max()), out=out, seq_order=order, products=prod, work=work)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(int(os.environ.get("AB_ROUNDS", 15))):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = sorted(ts)[len(ts) // 2]
    print(f"{label:28s} B={B:5d} T={T:6d}  {ms:.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s algorithmic", flush=True)
    return ms


quick = os.environ.get("AB_QUICK", "0") == "1"
for L in ((256, 512) if quick else (32, 64, 96, 128, 160, 192, 256, 320, 384, 448, 512)):
    run(torch.full((65536 // L,), L), f"uniform L={L}")
g = torch.Generator().manual_seed(1234)
lens = blair_sequence_lengths(256, g)
run(lens, "ragged users (bench)")
if quick:
    run(torch.cat([lens] * 4), "ragged users x 4 (tail amortised)")
    sys.exit(0)
run(lens[lens >= 384], "ragged, L >= 384 only")
run(lens[(lens >= 128) & (lens < 384)], "ragged, 128 <= L < 384")
run(lens[lens < 128], "ragged, L < 128")
