#!/usr/bin/env python3
"""Localize-and-Stitch and PCB task vectors at the model's full size (BUILD CONTAINER ONLY): tests/golden/g18_lns_pcb_fullsize.pt.

TEST INFRASTRUCTURE.  Companion of oracle/gen_golden_ties_fullsize.py (same seeded inputs): the reference's
``get_localize_and_stitch_vectors`` (8 models, density 0.05) and ``get_pcb_vectors`` (the first 4 models, density 0.2: its eight (n, d)
temporaries do not fit this container's memory with 8) on P = 124,645,632, reduced per model to the number of non-zeros, float64 sum and
absolute sum, and the values at 16,384 seeded positions.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

N_PCB, D_LNS, D_PCB = 4, 0.05, 0.2


def reduce(torch, out, pos):
    n = out.shape[0]
    return dict(nnz=[int((out[i] != 0).sum()) for i in range(n)], sum=[float(out[i].double().sum()) for i in range(n)],
                abs_sum=[float(out[i].double().abs().sum()) for i in range(n)], sample=out[:, pos].clone())


def main():
    import torch

    import gen_golden as GG
    import gen_golden_ties_fullsize as T

    torch.set_num_threads(8)
    GG.install_reference_importer()
    from rec_retrieval.merger.algorithms.localize_and_stitch import get_localize_and_stitch_vectors
    from rec_retrieval.merger.algorithms.pcb import get_pcb_vectors

    t0 = time.time()
    base, models = T.inputs(torch)
    pos = torch.randint(0, T.P, (T.N_SAMPLE,), generator=torch.Generator().manual_seed(T.SEED + 1))
    out = get_localize_and_stitch_vectors(base_model=base, models=models, density=D_LNS)
    lns = reduce(torch, out, pos)
    del out
    print(f"localize-and-stitch in {time.time() - t0:.0f}s", flush=True)
    out = get_pcb_vectors(base_model=base, models=models[:N_PCB], density=D_PCB)
    pcb = reduce(torch, out, pos)
    del out
    print(f"pcb in {time.time() - t0:.0f}s", flush=True)
    fx = dict(P=T.P, N=T.N, seed=T.SEED, sample_pos=pos, lns=lns, lns_density=D_LNS, pcb=pcb, pcb_models=N_PCB, pcb_density=D_PCB,
              versions=dict(torch=str(torch.__version__)))
    path = ROOT / "tests" / "golden" / "g18_lns_pcb_fullsize.pt"
    torch.save(fx, path)
    print("saved", path, path.stat().st_size, f"{time.time() - t0:.0f}s")


if __name__ == "__main__":
    main()
