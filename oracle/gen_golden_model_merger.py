#!/usr/bin/env python3
"""Fixed-weight ModelMerger merges beyond task_vector / linear (BUILD CONTAINER ONLY): tests/golden/g21_model_merger.pt.

TEST INFRASTRUCTURE.  The reference's ``ModelMerger(models, base_model).merge("ties" | "pcb" | "dare", weights, density=...)``
(rec_retrieval/merger/merger.py:46-93; algorithms/ties.py:74-83, pcb.py:60-72, dare.py:8-33) on the tiny state dicts of fixture g2
(its pretrained model and three fine-tuned ones).  "dare" draws ``torch.nn.functional.dropout`` masks from torch's global CPU generator:
the fixture records the seed set right before the call.
"""
from __future__ import annotations

import sys
from collections import OrderedDict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def main():
    import torch

    import gen_golden as GG

    GG.install_reference_importer()
    from rec_retrieval.merger.merger import ModelMerger

    g2 = torch.load(ROOT / "tests" / "golden" / "g2_merger.pt", weights_only=False)
    pre = OrderedDict((k, v) for k, v in g2["pretrain"].items())
    fts = [OrderedDict((k, ft[k]) for k in pre.keys()) for ft in g2["finetunes"]]
    flat = lambda sd: torch.cat([v.reshape(-1).float() for v in sd.values()])
    out = dict(weights=[0.5, 0.25, 0.7], cases={})
    for name, kw in (("ties", dict(density=0.2)), ("ties_dense", dict(density=0.9)), ("pcb", dict(density=0.2)), ("dare", dict(density=0.3))):
        mg = ModelMerger(models=[OrderedDict(f) for f in fts], base_model=OrderedDict(pre), align_key_order=False)
        seed = 1234
        torch.manual_seed(seed)
        merged = mg.merge(name.split("_")[0], list(out["weights"]), **kw)
        out["cases"][name] = dict(kwargs=kw, seed=seed, merged_flat=flat(merged).clone(), keys=list(merged.keys()))
        print(name, float(out["cases"][name]["merged_flat"].double().sum()))
    torch.save(out, ROOT / "tests" / "golden" / "g21_model_merger.pt")
    print("saved g21")


if __name__ == "__main__":
    main()
