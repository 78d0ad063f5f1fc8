#!/usr/bin/env python3
"""TIES task vectors at the model's full size (BUILD CONTAINER ONLY): tests/golden/g17_ties_fullsize.pt.

TEST INFRASTRUCTURE.  The reference's shipped recipe (scripts/3_mergerec/recformer_base_ties_layerwise.sh) pre-processes the task vectors with
TIES before it learns the coefficients.  g6 pins the device kernels against ``get_ties_vectors`` at P <= 20,000; this fixture runs the
reference's ``get_ties_vectors`` (merger/algorithms/ties.py) on the CPU at BLaIR-base's flat length P = 124,645,632 with 8 models and
density 0.2 -- 25 M survivors per model out of 125 M magnitudes, the regime the radix select and the sign election actually run in.
The (8, P) result (4 GB) is reduced to, per model: the number of non-zeros, float64 sum and absolute sum, and the values at 16,384 seeded
positions; inputs are regenerated from the seed by the test (one generator, base first, then the models in order).
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

P, N, DENSITY, SEED, N_SAMPLE = 124_645_632, 8, 0.2, 7100, 16384


def inputs(torch):
    g = torch.Generator().manual_seed(SEED)
    base = torch.randn(P, generator=g) * 0.02
    models = [base + torch.randn(P, generator=g) * 1e-3 for _ in range(N)]
    return base, models


def main():
    import torch

    import gen_golden as GG

    torch.set_num_threads(8)
    GG.install_reference_importer()
    from rec_retrieval.merger.algorithms.ties import get_ties_vectors

    t0 = time.time()
    base, models = inputs(torch)
    print(f"inputs in {time.time() - t0:.0f}s", flush=True)
    out = get_ties_vectors(base_model=base, models=models, density=DENSITY)
    print(f"get_ties_vectors in {time.time() - t0:.0f}s", tuple(out.shape), flush=True)
    pos = torch.randint(0, P, (N_SAMPLE,), generator=torch.Generator().manual_seed(SEED + 1))
    fx = dict(P=P, N=N, density=DENSITY, seed=SEED, sample_pos=pos, nnz=[int((out[i] != 0).sum()) for i in range(N)],
              sum=[float(out[i].double().sum()) for i in range(N)], abs_sum=[float(out[i].double().abs().sum()) for i in range(N)],
              sample=out[:, pos].clone(), versions=dict(torch=str(torch.__version__)))
    path = ROOT / "tests" / "golden" / "g17_ties_fullsize.pt"
    torch.save(fx, path)
    print("saved", path, path.stat().st_size, "nnz", fx["nnz"][:3], f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
