"""``ModelMerger``: the fixed-weight merges of rec_retrieval/merger/merger.py:10-110 on device-resident parameter arenas.

Same constructor and ``merge(merge_type, weights)`` surface as the reference; "task_vector" and "linear" run as one streaming HIP
pass each (``mr_merge_running_f32``: the reference's running sum in model order, every operation rounded on its own, so the result
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
        elif merge_type in ("ties", "dare", "pcb"):
            if self.base_model is None:
                raise ValueError(f"{merge_type.upper() if merge_type != 'dare' else 'DARE'} merge requires a base model.")
            raise NotImplementedError(f"ModelMerger.merge('{merge_type}') is not built: the TIES / PCB task vectors are built on the device by "
                                      "load_merging_module(MergeType.TIES | PCB, ...), the path the reference's scripts use; DARE draws torch "
                                      "dropout masks and has no entry script")
        else:
            raise ValueError(f"Merge type '{merge_type}' is not supported.")
        return self.layout.views(flat)  # unflatten_model (model_operations.py:66-90): named views, zero-copy
