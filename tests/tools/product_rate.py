#!/usr/bin/env python3
"""Steady-state rate of the drop-in evaluation loop (utils.test_model_on_dataloaders) on pre-tokenised synthetic domains of real sizes, with
a cProfile of the host side: how far is the product path (collation, coalescing, H2D, kernels, metric loops) from bench.py's kernel-side rate?
Usage: python tests/tools/product_rate.py [Pantry|Sports|...] [repeats]"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from mergerec_amd.data import load_domain  # noqa: E402
from mergerec_amd.evaluator import Evaluator  # noqa: E402
from mergerec_amd.module import ModelType, RecModule  # noqa: E402
from mergerec_amd.utils import test_model_on_dataloaders  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "Pantry"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    model = ModelType.BLAIR_BASE.value(model_kwargs={"init_seed": 3, "gemm_mode": "bf16x3"})
    module = RecModule(model=model, evaluator=Evaluator(["NDCG", "RECALL"], [1, 5, 10, 50]), similarity="cosine")
    t0 = time.perf_counter()
    dom = load_domain(f"synthetic:{name}")
    print(f"{name}: {dom.n_items} items, {dom.n_users} users (synthesised in {time.perf_counter() - t0:.1f}s)")
    for r in range(reps):
        pr = cProfile.Profile() if r == reps - 1 else None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pr:
            pr.enable()
        test_model_on_dataloaders(module, [dom.item_dataloader(32)], [dom.sequence_dataloader(32)], [name], precision="bf16-mixed")
        torch.cuda.synchronize()
        if pr:
            pr.disable()
        dt = time.perf_counter() - t0
        print(f"  run {r}: {dt:.3f}s  {dom.n_users / dt:.0f} users/s  ({(dom.n_users + dom.n_items) / dt:.0f} sequences+items/s)")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
