#!/usr/bin/env python3
"""Full-catalog scoring + top-50: selection inside the scoring kernel (csrc/score_fused.hip, the default when the score block is not
requested) (MR_SCORE_FUSED=1) against the scoring GEMM + topk_rows pair (MR_SCORE_FUSED=0), both through mr_score_topk_f32.
Usage: python tests/tools/score_fused_bench.py   (runs itself twice, once per path)"""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def run():
    import torch

    from mergerec_amd import ops

    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    tag = {"0": "staged", "1": "fused"}[os.environ["MR_SCORE_FUSED"]]
    for nU, M in ((256, 22855), (256, 4968), (32, 18357), (2048, 22855), (256, 114075)):
        base = torch.randn(1, 768, generator=g)  # embeddings of one domain cluster: cosine scores near 0.5-0.9, as real catalogs give
        U = torch.nn.functional.normalize(base + 0.7 * torch.randn(nU, 768, generator=g), dim=1).to(dev)
        E = torch.nn.functional.normalize(base + 0.7 * torch.randn(M, 768, generator=g), dim=1).to(dev)
        labels = torch.randint(0, M, (nU,), generator=g).to(dev)
        for _ in range(3):
            ops.score_topk(U, E, 50, labels, 20.0)
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.score_topk(U, E, 50, labels, 20.0)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        ws = ops._lib.load().mr_score_topk_ws_bytes_ex(nU, M, 768, 50)
        print(f"{tag:8s} users {nU:5d} x items {M:6d}: {ms:7.3f} ms  {2.0 * nU * M * 768 / ms / 1e9:6.1f} TFLOP/s  workspace {ws / 1e6:7.2f} MB")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        run()
    else:
        for env in ({"MR_SCORE_FUSED": "1"}, {"MR_SCORE_FUSED": "0"}):
            r = subprocess.run([sys.executable, __file__, "child"], env={**os.environ, **env}, capture_output=True, text=True)
            sys.stdout.write(r.stdout)
            if r.returncode:
                sys.stdout.write(r.stderr[-2000:])
