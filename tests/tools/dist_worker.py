#!/usr/bin/env python3
"""Worker of tests/test_dist_gpu.py: the drop-in merged-inference path (load_merging_module -> get_state_dict -> model ->
RecModule -> test_model) on one synthetic domain, task-wise and layer-wise, run either as one process or as N ranks under
torch.distributed.run (MERGEREC_DIST_BACKEND=gloo lets the ranks share one GPU).  Rank 0 saves everything the test compares."""
import os
import sys
from collections import OrderedDict
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))


def _digest(sd):
    """SHA-256 over the merged tensors' bytes (0.5 GB at true dimensions: the comparison is on the digest)"""
    import hashlib

    h = hashlib.sha256()
    for v in sd.values():
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def main(out_path: str):
    from mergerec_amd import parallel
    from mergerec_amd.data import load_domain
    from mergerec_amd.evaluator import Evaluator
    from mergerec_amd.merger import LearnType, MergeType, load_merging_module
    from mergerec_amd.module import ModelType, RecModule
    from mergerec_amd.utils import test_model_on_dataloaders

    rank, world = parallel.init_from_env()
    true_dims = os.environ.get("DIST_WORKER_TRUE_DIMS", "0") == "1"   # BLaIR-base as it is: a 124.6 M-parameter arena cut into slices
    over = dict(vocab=50265) if true_dims else dict(hidden=128, heads=2, layers=3, intermediate=256, vocab=400, max_pos=514)
    configs = ((("BLAIR_BASE", "LAYER_WISE", "TASK_VECTOR"),) if true_dims else
               (("BLAIR_BASE", "TASK_WISE", "TASK_VECTOR"), ("RECFORMER_BASE", "LAYER_WISE", "TASK_VECTOR"), ("BLAIR_BASE", "LAYER_WISE", "TIES")))
    out = {"world": world}
    for kind, learn, merge in configs:
        mk = {"init_seed": 21, "spec_overrides": dict(over)}
        model = ModelType[kind].value(model_kwargs=dict(mk))
        pre = OrderedDict((k, v.cpu().clone()) for k, v in model.state_dict().items())
        fts = []
        for i in range(3):
            g = torch.Generator().manual_seed(500 + i)
            fts.append(OrderedDict((k, v if k.endswith("position_ids") else v + (1e-3 if true_dims else 0.02) * torch.randn(v.shape, generator=g)) for k, v in pre.items()))
        mm = load_merging_module(MergeType[merge], LearnType[learn], model, pre, fts, set(), ties_density=0.3, disable_softmax=True)
        groups = list(mm.per_weights.keys())
        gg = torch.Generator().manual_seed(3)
        mm.load_weights_from_dict({"global_weights": {k: [1.0] for k in groups}, "global_biases": {k: [0.0] for k in groups},
                                   "per_weights": {k: (0.1 + 0.5 * torch.rand(3, generator=gg)).tolist() for k in groups}})
        sd = {k: v.detach() for k, v in mm.get_state_dict().items()}
        placement = "sliced" if mm.slice_plan is not None else "replicated"
        model2 = ModelType[kind].value(model_kwargs=dict(mk))
        model2.load_state_dict(sd)
        module = RecModule(model=model2, evaluator=Evaluator(["NDCG", "RECALL"], [1, 5, 10, 50]), similarity="cosine")
        dom = load_domain("synthetic:Toy:1500:700" if true_dims else "synthetic:Toy:333:301", kind="recformer" if kind.startswith("REC") else "roberta",
                          vocab=over["vocab"])
        metric_dict, metrics, scores, labels = test_model_on_dataloaders(
            module, [dom.item_dataloader(32)], [dom.sequence_dataloader(32)], ["Toy"], predictions_path=Path(out_path + f".pred_{kind}_{learn}_{merge}"))
        # the forward of the merging module itself (re-merges into the bound arena, then encodes): load_weights() under each placement
        batch = next(iter(dom.item_dataloader(16)))
        with torch.no_grad():
            cls = mm.forward(batch.to(model.device).items)
        out[f"{kind}/{learn}/{merge}"] = dict(
            placement=placement, merged=(_digest(sd) if true_dims else torch.cat([v.reshape(-1) for v in sd.values()]).cpu()), item_embeddings=module.item_embeddings.detach().cpu(),
            user_embeddings=module.eval_user_embeddings, topk=module.eval_topk_indices, labels=module.eval_labels, metrics=metrics[0],
            scores=scores[0], mm_forward_cls=cls.cpu())
    if rank == 0:
        torch.save(out, out_path)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
