"""Token-sized training products: mr_gemm_tile_f32 (csrc/gemm_train.hip; no transposes, no split-K) against the r03 route (transpose + split-K
NT kernel + reduce), per product of one encoder layer at T tokens.   PYTHONPATH=. python tools/gemm_tile_bench.py [T] [hidden]"""
import sys

import torch

from mergerec_amd import ops

dev = torch.device("cuda:0")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 602
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
di = 4 * d
g = torch.Generator().manual_seed(0)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3  # us


rows = []
x, xi = torch.randn(T, d, device=dev), torch.randn(T, di, device=dev)
W = {n: torch.randn(o, i, device=dev) * 0.02 for n, (o, i) in dict(q=(d, d), k=(d, d), v=(d, d), o=(d, d), w1=(di, d), w2=(d, di)).items()}
dy_d, dy_di, dy_3d = torch.randn(T, d, device=dev), torch.randn(T, di, device=dev), torch.randn(T, 3 * d, device=dev)
cases = [
    ("fwd qkv  (T,3d,d)", 2.0 * T * 3 * d * d, lambda bn: ops.gemm_tile(x, [W["q"], W["k"], W["v"]], bn=bn), lambda: [ops.gemm_nt_train(x, W[n]) for n in "qkv"]),
    ("fwd out  (T,d,d)", 2.0 * T * d * d, lambda bn: ops.gemm_tile(x, [W["o"]], bn=bn), lambda: ops.gemm_nt_train(x, W["o"])),
    ("fwd ffn1 (T,4d,d)", 2.0 * T * di * d, lambda bn: ops.gemm_tile(x, [W["w1"]], bn=bn), lambda: ops.gemm_nt_train(x, W["w1"])),
    ("fwd ffn2 (T,d,4d)", 2.0 * T * di * d, lambda bn: ops.gemm_tile(xi, [W["w2"]], bn=bn), lambda: ops.gemm_nt_train(xi, W["w2"])),
    ("dX  qkv  (T,d,3d)", 2.0 * T * 3 * d * d, lambda bn: ops.gemm_tile(dy_3d, [W["q"], W["k"], W["v"]], trans_b=True, bn=bn),
     lambda: ops.gemm_nt_train(dy_3d, torch.cat([ops.transpose_pad(W[n]) for n in "qkv"], dim=1))),
    ("dX  ffn1 (T,d,4d)", 2.0 * T * di * d, lambda bn: ops.gemm_tile(dy_di, [W["w1"]], trans_b=True, bn=bn), lambda: ops.gemm_nt_train(dy_di, ops.transpose_pad(W["w1"]))),
    ("dX  ffn2 (T,4d,d)", 2.0 * T * di * d, lambda bn: ops.gemm_tile(dy_d, [W["w2"]], trans_b=True, bn=bn), lambda: ops.gemm_nt_train(dy_d, ops.transpose_pad(W["w2"]))),
    ("dW  out  (d,d,T)", 2.0 * T * d * d, lambda bn: ops.gemm_tile(dy_d, [x], trans_a=True, trans_b=True, bn=bn),
     lambda: ops.gemm_nt_train(ops.transpose_pad(dy_d), ops.transpose_pad(x))),
    ("dW  ffn1 (4d,d,T)", 2.0 * T * di * d, lambda bn: ops.gemm_tile(dy_di, [x], trans_a=True, trans_b=True, bn=bn),
     lambda: ops.gemm_nt_train(ops.transpose_pad(dy_di), ops.transpose_pad(x))),
    ("dW  ffn2 (d,4d,T)", 2.0 * T * di * d, lambda bn: ops.gemm_tile(dy_d, [xi], trans_a=True, trans_b=True, bn=bn),
     lambda: ops.gemm_nt_train(ops.transpose_pad(dy_d), ops.transpose_pad(xi))),
    ("dW  qkv  (3d,d,T)", 2.0 * T * 3 * d * d, lambda bn: ops.gemm_tile(dy_3d, [x], trans_a=True, trans_b=True, bn=bn),
     lambda: ops.gemm_nt_train(ops.transpose_pad(dy_3d), ops.transpose_pad(x))),
]
tot = dict(old=0.0, new=0.0, ideal=0.0)
print(f"T = {T}, hidden = {d}: microseconds per product (fp32 MFMA peak 157.3 TFLOP/s)")
for name, flops, new, old in cases:
    t_old, t32, t64, t0 = timed(old), timed(lambda: new(32)), timed(lambda: new(64)), timed(lambda: new(0))
    ideal = flops / 157.3e12 * 1e6
    tot["old"] += t_old; tot["new"] += t0; tot["ideal"] += ideal
    print(f"  {name:20s} r03 route {t_old:7.1f}   tile32 {t32:7.1f}  tile64 {t64:7.1f}  auto {t0:7.1f}   at peak {ideal:6.1f}   auto = {ideal / t0:.2f} of peak", flush=True)
print(f"  one layer: r03 {tot['old']:.0f} us, tile kernel {tot['new']:.0f} us, at peak {tot['ideal']:.0f} us")
