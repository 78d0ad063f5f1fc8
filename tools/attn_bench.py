"""Micro-benchmark of mr_attn_f32 on an Amazon-shaped batch (HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergerec_amd import ops
from mergerec_amd.synthetic import blair_sequence_lengths, blair_item_lengths

dev = "cuda:0"
g = torch.Generator().manual_seed(1234)
mode = os.environ.get("AB_MODE", "mixed")
if mode == "uniform":
    lens = torch.full((256,), int(os.environ.get("AB_L", 256)))
else:
    lens = torch.cat([blair_item_lengths(128, g), blair_sequence_lengths(256, g)])
B, H = lens.numel(), 12
cu = torch.zeros(B + 1, dtype=torch.int32); cu[1:] = lens.cumsum(0)
T = int(cu[-1])
qkv = torch.randn(T, 3 * H * 64, device=dev)
cu_d = cu.to(dev)
out = torch.empty(T, H * 64, device=dev)
flops = 4.0 * 768 * float((lens.double() ** 2).sum())
pad = lambda x, m: (x + m - 1) // m * m
work = 4.0 * 768 * float((pad(lens, 32).double() ** 2).sum())
prod = int(os.environ.get("AB_PRODUCTS", "0"))
order = torch.argsort(lens, descending=True, stable=True).to(torch.int32).to(dev) if os.environ.get("AB_ORDER", "1") == "1" else None
for _ in range(3):
    ops.attention(qkv, cu_d, B, H, int(lens.max()), out=out, seq_order=order, products=prod)
torch.cuda.synchronize()
ts = []
for _ in range(int(os.environ.get("AB_ROUNDS", 10))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.attention(qkv, cu_d, B, H, int(lens.max()), out=out, seq_order=order, products=prod); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = sorted(ts)[len(ts) // 2]
print(f"products={prod} {mode}: B={B} T={T} max_len={int(lens.max())}  {ms:.3f} ms  algorithmic {flops/ms/1e9:.1f} TFLOP/s  tile-padded work {work/ms/1e9:.1f} TFLOP/s")
