"""``ModelMerger``: the fixed-weight merges of rec_retrieval/merger/merger.py:10-110 on device-resident parameter arenas.

Same constructor and ``merge(merge_type, weights, **kwargs)`` surface as the reference ("linear", "task_vector", "ties", "dare", "pcb");
"task_vector" and "linear" run as one streaming HIP pass each (``mr_merge_running_f32``: the reference's running sum in model order, every operation rounded on its own, so the result
is bit-for-bit the reference's).  The learnable-alpha path (``load_merging_module``) is the one merge_test.py uses; this class is
the reference's stand-alone merger kept for callers that hold N state dicts and a weight list."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Union

import torch

from .. import ops
from ..engine import ArenaLayout

StateDict = Dict[str, torch.Tensor]


def check_model_shape(models: Sequence[StateDict], base_model: Optional[StateDict] = None) -> None:
    """utils/model_operations.py:15-44: identical key sets and shapes, or AssertionError with the reference's messages."""
    keys = set(models[0].keys())
    for m in models:
        assert set(m.keys()) == keys, "Models have different architectures."
    for name in keys:
        for m in models[1:]:
            assert m[name].shape == models[0][name].shape, "Models have different shapes."
    if base_model is not None:
        assert set(base_model.keys()) == keys, "Base model has different architecture from the others."
        for name in keys:
            assert base_model[name].shape == models[0][name].shape, "Base model has different shapes."


class ModelMerger:
    def __init__(self, models: Sequence[StateDict], base_model: Optional[StateDict] = None, align_key_order: bool = True,
                 device: Optional[torch.device] = None):
        check_model_shape(models, base_model)
        models = list(models)
        if align_key_order:  # align_dict_key_order: the sorted key order of the first model (model_operations.py:93-136)
            order = sorted(models[0].keys())
            models = [OrderedDict((k, m[k]) for k in order) for m in models]
            base_model = None if base_model is None else OrderedDict((k, base_model[k]) for k in order)
        else:
            ref = list(models[0].keys())
            others = models[1:] + ([base_model] if base_model is not None else [])
            assert all(list(m.keys()) == ref for m in others), "Model keys are not aligned."
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.layout = ArenaLayout(OrderedDict((k, tuple(v.shape)) for k, v in models[0].items()))
        self.shape_dict = OrderedDict((k, torch.Size(s)) for k, s in self.layout.shapes.items())
        # (N, P_pad) parameters in arena layout; with no base model the FIRST model is the head of the list, as in merger.py:33-41
        self.models = torch.stack([self.layout.pack(m, device) for m in models])
        self.base_model = None if base_model is None else self.layout.pack(base_model, device)

    @torch.no_grad()
    def merge(self, merge_type: str, weights: Union[Sequence[float], float], **kwargs) -> StateDict:
        n = self.models.shape[0]
        if isinstance(weights, float):
            weights = [weights] * n
        elif not (isinstance(weights, list) and all(isinstance(w, float) for w in weights)):
            raise ValueError("Weights should be a float or a list of floats.")
        assert len(weights) == n, "Number of models and weights should match."
        w = torch.tensor(weights, dtype=torch.float32, device=self.models.device)  # python floats times fp32 tensors: fp32 products
        if merge_type == "linear":
            flat = ops.merge_running(None, self.models, w)
        elif merge_type == "task_vector":
            if self.base_model is None:
                raise ValueError("Task vector merge requires a base model.")
            flat = ops.merge_running(self.base_model, self.models, w)
        elif merge_type == "ties":
            if self.base_model is None:
                raise ValueError("TIES merge requires a base model.")
            flat = self._merge_ties(weights, **kwargs)
        elif merge_type == "dare":
            if self.base_model is None:
                raise ValueError("DARE merge requires a base model.")
            flat = self._merge_dare(weights, **kwargs)
        elif merge_type == "pcb":
            if self.base_model is None:
                raise ValueError("PCB merge requires a base model.")
            flat = self._merge_pcb(w, **kwargs)
        else:
            raise ValueError(f"Merge type '{merge_type}' is not supported.")
        return self.layout.views(flat)  # unflatten_model (model_operations.py:66-90): named views, zero-copy

    # ---- the three merges that pre-process the task vectors (every one needs a base model; `density` as in the reference's signatures)
    def _scatter_compact(self, compact: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """(P,) in the reference's flat order -> arena layout (pads stay as they are in ``out``)"""
        off = 0
        for k, shp in self.layout.shapes.items():
            cnt = 1
            for x_ in shp:
                cnt *= x_
            out[self.layout.offsets[k]: self.layout.offsets[k] + cnt] = compact[off: off + cnt]
            off += cnt
        return out

    def _merge_ties(self, weights: List[float], density: float, **__) -> torch.Tensor:
        """algorithms/ties.py:74-83 (merge_ties): base + sum_i top-k_{|.|}(w_i (theta_i - base)), k = int(density * numel) -- no sign
        election on this path (that is get_ties_vectors, the learnable module's pre-processing).  The sum runs in model order."""
        n, base = self.models.shape[0], self.base_model
        k = int(density * self.layout.numel)
        sparse = torch.empty_like(self.models)
        for i in range(n):
            ops.task_vector(self.models[i], base, out=sparse[i])
            sparse[i].mul_(weights[i])                       # `update *= weights[i]` (ties.py:22-23): one fp32 product
            ops.abs_topk_mask(sparse[i], k, out=sparse[i])   # ties at the k-th magnitude: lowest indices (torch.topk leaves them unspecified)
        delta = ops.merge_running(None, sparse, torch.ones(n, dtype=torch.float32, device=sparse.device))  # torch.sum(dim=0): rows in order
        return base + delta

    def _merge_pcb(self, w: torch.Tensor, density: float = 0.2, **__) -> torch.Tensor:
        """algorithms/pcb.py:60-72 (merge_pcb): merged = base; merged += w_i * pcb_i in model order.  The PCB vectors are order statistics
        of each task vector, so they are computed on the compact (unpadded) vectors and scattered back."""
        n, base = self.models.shape[0], self.base_model
        tv = torch.stack([self.layout.compact(ops.task_vector(self.models[i], base)) for i in range(n)]).contiguous()
        pcb = ops.pcb_vectors(tv, density)
        arena = torch.zeros_like(self.models)
        for i in range(n):
            self._scatter_compact(pcb[i], arena[i])
        # running sum `merged += w_i * v_i` from merged = base: the "task_vector" kernel on theta_i := base + v_i would round v_i away, so the
        # sum runs as a linear running sum over [base, v_0, ...] with weights [1, w_0, ...] (1 * base is exact, 0 + base is exact)
        rows = torch.cat([base.unsqueeze(0), arena])
        return ops.merge_running(None, rows, torch.cat([torch.ones(1, dtype=torch.float32, device=w.device), w]))

    def _merge_dare(self, weights: List[float], density: float, **__) -> torch.Tensor:
        """algorithms/dare.py:8-33 (merge_dare): merged += dropout(w_i (theta_i - base), p=density, training=True) in model order.  The masks
        come from torch's global CPU generator exactly as in the reference (one ``dropout`` call per model over the P flat elements, in the
        reference's flat order), so a run seeded like the reference reproduces it; the arithmetic runs on the device."""
        from torch.nn.functional import dropout

        n, base = self.models.shape[0], self.base_model
        P = self.layout.numel
        upd = torch.empty_like(self.models)
        for i in range(n):
            ops.task_vector(self.models[i], base, out=upd[i])
            upd[i].mul_(weights[i])
            keep = dropout(torch.ones(P, dtype=torch.float32), p=density, training=True)       # 0 or 1 / (1 - p), the reference's draw
            scale = self._scatter_compact(keep.to(upd.device), torch.zeros_like(base))
            upd[i].mul_(scale)                                # x * (1 / (1 - p)) or x * 0: what dropout computes per element
        rows = torch.cat([base.unsqueeze(0), upd])
        return ops.merge_running(None, rows, torch.ones(n + 1, dtype=torch.float32, device=upd.device))
