"""Host-side mirror of rec_retrieval/merger/weight_learning (the learnable-alpha merge module).

Same names, arguments and error behaviour as the reference; the arithmetic runs in the HIP kernels
mr_task_vector_f32 / mr_merge_nway_f32 on device-resident arena buffers (mergerec_amd/engine.py).

  load_merging_module            <- weight_learning/module/_factory.py:27-127
  TaskVectorMergingModuleBase    <- weight_learning/module/_base.py:10-89
  TaskVectorMergingModuleTaskWise  <- weight_learning/module/task_wise.py:12-62
  TaskVectorMergingModuleLayerWise <- weight_learning/module/layer_wise.py:37-95
"""
from __future__ import annotations

from collections import OrderedDict
import weakref
from typing import Dict, List, Optional

import torch
from torch import nn

import os

from .. import ops, parallel
from ..engine import ArenaLayout
from .enums import LearnType, MergeType

StateDict = Dict[str, torch.Tensor]


class MergeOverlap:
    """Chunk plan of ONE alpha-learning step: the merge and the alpha-gradient contraction are parameter-sized HBM streams, the encoder's
    600-token products beside them barely touch HBM -- so both streams run in arena ranges (the embeddings, each encoder layer, the
    pooler: ``ArenaLayout.group_segments``) on a SECOND stream, under the encoder's kernels.  Forward: every range is merged in arena
    order and the training graph waits for a layer's event just before that layer's first product.  Backward: a layer's range is
    contracted with the task vectors as soon as the layer's gradients are complete (the weight gradients already run on that stream);
    only the embedding range -- complete last -- is left at the end.  Same kernels on sub-ranges: the merged parameters are bit-identical;
    d alpha is the same sums grouped per range (ranges in arena order: deterministic)."""

    def __init__(self, layout: ArenaLayout, stream: "torch.cuda.Stream", word_key: Optional[str] = None, tok_word: Optional[torch.Tensor] = None):
        groups, seg_off, seg_gid = layout.group_segments()
        self.bounds = [int(v) for v in seg_off.tolist()]
        self.keys = [groups[g] for g in seg_gid]            # "others" (embeddings), "0", "1", ..., "others" (pooler)
        self.stream = stream
        self.events = [None] * len(self.keys)
        self.partials = [None] * len(self.keys)
        self.hold = []                                        # what the side stream reads, until the main stream has joined it
        self.tv = self.g_ptr = None
        # The word-embedding table is a third (BLaIR-base) / an eighth (Recformer-large) of the parameters and a step reads ~600 of its
        # 50,265 rows: with the batch's row list the table is merged and contracted in those rows only (``word``: range, rows, width, chunk).
        self.word, self.tok_word = None, tok_word
        if word_key is not None and tok_word is not None and word_key in layout.offsets and len(layout.shapes[word_key]) == 2:
            w0, (V, d) = int(layout.offsets[word_key]), layout.shapes[word_key]
            c = max(i for i in range(len(self.keys)) if self.bounds[i] <= w0)
            if d % 4 == 0 and w0 % 4 == 0 and w0 + V * d <= self.bounds[c + 1]:
                self.word = (w0, int(V), int(d), c)

    def _dense_ranges(self, c: int):
        """the arena ranges of chunk c that are streamed whole (everything but the word table when its rows are listed)"""
        b, e = self.bounds[c], self.bounds[c + 1]
        if self.word is None or self.word[3] != c:
            return [(b, e)]
        w0, V, d, _ = self.word
        return [r for r in ((b, w0), (w0 + V * d, e)) if r[1] > r[0]]

    # ---- forward
    def merge(self, base, tv, alpha, seg_off, out):
        main = torch.cuda.current_stream(base.device)
        self.stream.wait_stream(main)                         # alpha's two torch ops; the previous readers of the recycled `out` block
        self.hold += [alpha, out]
        self.tv = tv
        with torch.cuda.stream(self.stream):
            for c in range(len(self.keys)):
                if self.word is not None and self.word[3] == c:   # the batch's rows of the word table (the other rows stay unwritten: nothing reads them)
                    w0, V, d, _ = self.word
                    a_row = alpha.view(-1, tv.shape[0])[c if seg_off is not None else 0]
                    ops.merge_rows(base, tv, a_row, self.tok_word, V, d, w0, out)
                for b, e in self._dense_ranges(c):
                    ops.merge_nway(base, tv, alpha, seg_off, out=out, p_begin=b, p_count=e - b)
                self.events[c] = torch.cuda.Event()
                self.events[c].record(self.stream)
        return out

    def wait(self, key: Optional[str] = None, first_only: bool = False):
        """the main stream waits for the merged range(s) of group ``key`` (None: everything not yet waited for)"""
        main = torch.cuda.current_stream()
        for c, k in enumerate(self.keys):
            if self.events[c] is not None and (key is None or k == key):
                main.wait_event(self.events[c])
                self.events[c] = None
                if first_only:
                    return

    # ---- backward
    def contract(self, key: str, g_flat: torch.Tensor, d_emb: Optional[torch.Tensor] = None):
        """d alpha of group ``key``'s range(s), on the side stream, ordered after everything issued so far on both streams.  ``d_emb`` (T, d):
        d loss / d (embedding sum) per token -- the word table's share is then sum_t <tau_i[row(t)], d_emb[t]> over the batch's rows (what the
        dense contraction of the scatter-added table gives, without streaming the table)."""
        main = torch.cuda.current_stream(g_flat.device)
        self.stream.wait_stream(main)
        self.g_ptr = g_flat.data_ptr()
        self.hold.append(g_flat)
        with torch.cuda.stream(self.stream):
            for c, k in enumerate(self.keys):
                if k != key or self.partials[c] is not None:
                    continue
                sparse = self.word is not None and self.word[3] == c and d_emb is not None
                ranges = self._dense_ranges(c) if sparse else [(self.bounds[c], self.bounds[c + 1])]
                parts = [ops.merge_bwd_alpha(self.tv, g_flat, p_begin=b, p_count=e - b) for b, e in ranges]
                if sparse:
                    w0, V, d, _ = self.word
                    self.hold.append(d_emb)
                    rows = torch.stack([ops.gather_rows(self.tv[i, w0:w0 + V * d].view(V, d), self.tok_word) for i in range(self.tv.shape[0])])
                    parts.append(ops.merge_bwd_alpha(rows.view(rows.shape[0], -1), d_emb.reshape(-1)))
                total = parts[0]
                for q in parts[1:]:
                    total = total + q
                self.partials[c] = total

    def dalpha(self, g: torch.Tensor, n_segments: int) -> Optional[torch.Tensor]:
        """(S, N) from the per-range contractions, or None when they do not cover this gradient vector (the caller contracts in one launch)"""
        if self.g_ptr != g.data_ptr() or any(p is None for p in self.partials):
            return None
        torch.cuda.current_stream(g.device).wait_stream(self.stream)
        for t in self.partials:
            t.record_stream(torch.cuda.current_stream(g.device))
        rows = torch.cat(self.partials, dim=0)               # (ranges, N), arena order
        self.hold.clear()
        return rows if n_segments == rows.shape[0] else rows.sum(dim=0, keepdim=True)


class _MergeFunction(torch.autograd.Function):
    """a20: the merge as an autograd node.  forward = mr_merge_nway_f32, backward = mr_merge_bwd_alpha_f32
    (dalpha[s, i] = <tau_i[segment s], dL/dtheta[segment s]>), i.e. the gradient the reference obtains by
    differentiating task_wise.py:43-47 / layer_wise.py:75-81 inside merge_train.py's training step.
    ``plan`` (a MergeOverlap): both streams in arena ranges on a second stream, under the encoder's kernels."""

    @staticmethod
    def forward(ctx, alpha, base, tv, seg_off, out, plan=None):
        ctx.tv, ctx.seg_off, ctx.plan = tv, seg_off, plan
        a = alpha.detach().contiguous()
        if plan is not None:
            return plan.merge(base, tv, a, seg_off, out)
        ops.merge_nway(base, tv, a, seg_off, out=out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        g = grad_out.contiguous()
        S = 1 if ctx.seg_off is None else ctx.seg_off.numel() - 1
        dalpha = ctx.plan.dalpha(g, S) if ctx.plan is not None else None
        if dalpha is None:
            if ctx.plan is not None:
                torch.cuda.current_stream(g.device).wait_stream(ctx.plan.stream)
            dalpha = ops.merge_bwd_alpha(ctx.tv, g, ctx.seg_off)
        return dalpha, None, None, None, None, None


def _check_isinstance_state_dict(t):
    """_factory.py:17-25."""
    if not isinstance(t, dict):
        raise ValueError(f"Expected a state dict, got {type(t)}")
    for k, v in t.items():
        if not isinstance(k, str):
            raise ValueError(f"Expected a string key, got {type(k)}")
        if not isinstance(v, torch.Tensor):
            raise ValueError(f"Expected a tensor value, got {type(v)}")


class TaskVectorMergingModuleBase(nn.Module):
    """_base.py:10-89.  ``base_model_tensor`` / ``task_vectors_tensor`` are device buffers in ARENA layout
    ((P_pad,) / (N, P_pad): each tensor 64-float aligned, pads zero); ``compact_*`` give the reference's
    contiguous (P,) / (N, P) forms."""

    def __init__(self, base_model_tensor, task_vectors_tensor, model_without_params, layout: ArenaLayout, disable_softmax: bool = False,
                 slice_plan: Optional["parallel.SlicePlan"] = None):
        super().__init__()
        self.model = model_without_params
        self.layout = layout
        self.shape_dict = OrderedDict((k, torch.Size(s)) for k, s in layout.shapes.items())
        self.disable_softmax = disable_softmax
        self.base_model_tensor = nn.Parameter(base_model_tensor, requires_grad=False)
        self.task_vectors_tensor = nn.Parameter(task_vectors_tensor, requires_grad=False)
        self.global_weights = nn.ParameterDict()
        self.global_biases = nn.ParameterDict()
        self.per_weights = nn.ParameterDict()
        # "sliced" placement (several ranks): this rank holds elements [lo, hi) of the base vector and of every task vector, merges
        # that slice, and ONE all-gather assembles the arena on every rank (parallel.sharded_merge)
        self.slice_plan = slice_plan
        self._slice = slice_plan.bounds(parallel.world()[0]) if slice_plan is not None else None
        # the parameter arena the (param-less) model reads: merged weights are written here in place (a6)
        arena_len = layout.padded_numel if slice_plan is None else slice_plan.padded
        self._arena = torch.zeros(arena_len, dtype=torch.float32, device=base_model_tensor.device)
        self._merged = self._arena[: layout.padded_numel]
        self._scratch = None
        self._seg_off: Optional[torch.Tensor] = None  # device int64 (S+1) or None for one segment
        self._seg_gid: List[int] = [0]
        self._groups: List[str] = ["all"]
        if hasattr(self.model, "bind_arena"):
            self.model.bind_arena(layout, self._merged)

    # -- reference API ---------------------------------------------------------------------------
    def trainable_parameters(self, freeze_global_weight=False, freeze_global_bias=False, freeze_per_weight=False):
        params = []
        if not freeze_global_weight:
            params.extend(self.global_weights.parameters())
        if not freeze_global_bias:
            params.extend(self.global_biases.parameters())
        if not freeze_per_weight:
            params.extend(self.per_weights.parameters())
        return params

    def serialize_weights(self):
        return {
            "global_weights": {k: v.tolist() for k, v in self.global_weights.items()},
            "global_biases": {k: v.tolist() for k, v in self.global_biases.items()},
            "per_weights": {k: v.tolist() for k, v in self.per_weights.items()},
        }

    @torch.no_grad()
    def load_weights_from_dict(self, weights: Dict[str, Dict[str, List[float]]]):
        dev = self.base_model_tensor.device
        for name, table, truncate in (("global_weights", self.global_weights, False), ("global_biases", self.global_biases, False),
                                      ("per_weights", self.per_weights, True)):
            for k, v in weights[name].items():
                assert k in table, f"Key '{k}' not found in {name}."
                v = torch.tensor(v)
                if truncate:
                    v = v[: table[k].numel()]  # _base.py:72
                assert v.shape == table[k].shape, f"Shape mismatch for key '{k}', ({v.shape} != {table[k].shape})"
                table[k].data = v.to(dev)
        if getattr(self, "_pipe", None) is not None and self._pipe["pending"] is not None:
            pend = self._pipe["pending"]  # a prefetched merge speculated on the old coefficients: its arena is the spare again
            self._pipe["spare"], self._pipe["spare_ws"], self._pipe["pending"] = pend["arena"], pend["ws"], None

    def forward(self, batch):
        """_base.py:78-81.  Under autograd (alpha requires grad and grad mode is on) the merge AND the encoder are differentiable:
        merged_params() -> RobertaTrainGraph, whose backward hands d loss / d merged parameters to the alpha-gradient kernel.
        Otherwise the merged weights are written into the model's arena and the inference path runs."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return self.forward_with_grad(batch)
        self.load_weights()
        return self.model(batch)

    def encode_normalized(self, batch, normalize: bool, lens=None, validate=True) -> torch.Tensor:
        """The evaluation loops' form of ``forward`` (RecModule._encode): re-merge, then the model's encode with the L2 normalisation
        fused into the pooling kernel and the input checks deferred to ``check_inputs()``."""
        self.load_weights()
        return self.model.encode_normalized(batch, normalize=normalize, lens=lens, validate=validate)

    def check_inputs(self) -> None:
        """Deferred input-contract check of the wrapped encoder (``BaseEncoderModel.check_inputs``): raises ``engine.InputError``."""
        self.model.check_inputs()

    # arithmetic of the training graph's products: "f32" exact-fp32 MFMA on the token-sized tile kernel; "bf16x3" split-precision MFMA on the
    # library's 256-wide tiles (the weights are re-split every step); "auto" = by the batch's token count -- the tile kernel below
    # AUTO_SPLIT_TOKENS tokens, the split graph from there on (measured at BLaIR-base x 8: 602 tokens 8.4 vs 11.5 ms, 1,202 tokens 12.5 vs
    # 12.6, 1,859 tokens 17.7 vs 14.0, 4,782 tokens 39.9 vs 20.5).  DistillTrainer sets "auto" for the reduced-precision flags
    # (bf16-mixed, the reference's default, 16-mixed, ...) and "f32" for 32-true.
    train_mode = "f32"
    AUTO_SPLIT_TOKENS = 1100  # Recformer-large x 8: 603 tokens 30.9 vs 36.4 ms, 1,159 tokens 44.5 vs 41.3

    def forward_with_grad(self, batch):
        from ..engine_train import RobertaTrainGraph, SplitWeights, encode_with_grad

        if getattr(self.model, "pooling_method", "cls") != "cls":
            raise NotImplementedError(f"the training graph pools the CLS row; pooling_method={self.model.pooling_method!r} is an inference option here")
        if self.slice_plan is not None:
            raise RuntimeError("alpha learning needs every task vector on every rank (d loss / d alpha contracts the full gradient with "
                               "each of them): build the module with placement='replicated' (merge_train.py does)")
        from .. import engine_train as _ET

        pb = self.model.runner.pack(batch, self.base_model_tensor.device)
        mode = self.train_mode
        if mode == "auto":
            mode = "bf16x3" if pb.T >= self.AUTO_SPLIT_TOKENS else "f32"
        if self.model.spec.hidden % 128:
            mode = "f32"
        plan = None
        if mode == "f32" and _ET._TILE and _ET._MERGE_OVERLAP:   # the tile graph knows the per-layer hooks; MR_TRAIN_MERGE_OVERLAP=0: one launch each
            if getattr(self, "_overlap_stream", None) is None:
                self._overlap_stream = torch.cuda.Stream(device=self.base_model_tensor.device)
            rows = _ET._SPARSE_WORD_ROWS
            plan = MergeOverlap(self.layout, self._overlap_stream, self.model.runner.prefix + "embeddings.word_embeddings.weight" if rows else None,
                                pb.tok_word if rows else None)
        # the merged vector and the gradient arena of the step live in two persistent buffers of this module (their 200-odd named views
        # are built once): every step writes both in full before reading them, on streams ordered behind the previous step's readers
        prev = getattr(self, "_train_graph", None)
        prev = prev() if prev is not None else None   # weak: a graph that was dropped without a backward frees the buffers again
        if prev is not None and prev._saved is not None:
            # a second differentiable forward before the first one's backward (two views of a batch, accumulated micro-batches, ...): the
            # first graph still reads the persistent buffers -- this call gets vectors of its own
            cache = (None, None)
            merged = self.merged_params(plan)
        else:
            cache = getattr(self, "_train_cache", None)
            if cache is None:
                pbuf, gbuf = torch.empty_like(self._merged), torch.empty_like(self._merged)
                cache = self._train_cache = ((pbuf, self.layout.views(pbuf)), (gbuf, self.layout.views(gbuf)))
            merged = self.merged_params(plan, out=cache[0][0].detach())
        sw = None
        if mode == "bf16x3":  # the merged weights are new every step: re-split them (and their transposes) from the merged arena
            sw = getattr(self, "_split_weights", None)
            if sw is None:
                sw = self._split_weights = SplitWeights(self.model.spec, self.layout, self.model.runner.prefix, self.base_model_tensor.device)
            sw.refresh(merged.detach())
        # dropout as the reference's functional call of the HF model in train() mode has it (merge_train.py:178-196): active iff THIS module
        # is in training mode, rates / seed / counter from the wrapped model
        graph = RobertaTrainGraph(self.model.spec, self.layout, prefix=self.model.runner.prefix, mode=mode, split_weights=sw,
                                  dropout=self.model.next_dropout(training=self.training))
        graph.overlap = plan
        graph.param_cache, graph.grad_cache = cache
        if cache[0] is not None:
            self._train_graph = weakref.ref(graph)
        return encode_with_grad(graph, merged, pb)

    # -- merge -----------------------------------------------------------------------------------
    def effective_alpha(self) -> torch.Tensor:
        """(S, N) device table: alpha = gw * (softmax?)(per) + gb per group (task_wise.py:37-42,
        layer_wise.py:67-73), expanded to the arena's segments.  Two tiny torch ops (mul, add), each
        rounded separately like the reference."""
        rows = {}
        for key in self._groups:
            per = self.per_weights[key]
            if not self.disable_softmax:
                per = torch.softmax(per, dim=0)
            rows[key] = self.global_weights[key] * per + self.global_biases[key]
        return torch.stack([rows[self._groups[g]] for g in self._seg_gid]).contiguous()

    @torch.no_grad()
    def _merge_task_vectors(self, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Writes the merged parameters into ``out`` (default: the model's arena) and returns the (P_pad,) vector."""
        alpha = self.effective_alpha().detach()
        if self.slice_plan is None:
            out = self._merged if out is None else out
            return ops.merge_nway(self.base_model_tensor.data, self.task_vectors_tensor.data, alpha, self._seg_off, out=out)
        arena = self._arena if out is None else out
        if arena.numel() != self.slice_plan.padded:
            raise ValueError("sliced merge writes a whole arena of SlicePlan.padded elements")
        if self._scratch is None:
            self._scratch = torch.empty(self.slice_plan.slice_len, dtype=torch.float32, device=arena.device)

        def merge_slice(p_begin, p_count, out_slice):
            ops.merge_nway(self.base_model_tensor.data, self.task_vectors_tensor.data, alpha, self._seg_off, out=out_slice,
                           p_begin=p_begin, p_count=p_count, out_is_slice=True, operands_are_slices=True)

        parallel.sharded_merge(merge_slice, arena, self.slice_plan, self._scratch)
        return arena[: self.layout.padded_numel]

    def merged_params(self, plan: Optional[MergeOverlap] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Differentiable merge: returns the merged arena vector with an autograd edge back to
        global_weights / global_biases / per_weights (the backward runs the HIP alpha-gradient kernel).  With ``plan`` the vector is
        written range by range on the plan's stream: its consumer waits per range (``MergeOverlap.wait``)."""
        return _MergeFunction.apply(self.effective_alpha(), self.base_model_tensor.data, self.task_vectors_tensor.data,
                                    self._seg_off, torch.empty_like(self._merged) if out is None else out, plan)

    # -- pipelined re-merge (evaluation loops that re-merge per step, as the reference does on every forward) ---------------------------
    def pipeline_merges(self, on: bool = True):
        """Double-buffer the merged arena: while the encoder of step s reads arena A, the merge (+ weight split) of step s + 1 runs on a
        SECOND HIP stream into arena B, speculating that alpha stays what it is now; the next ``load_weights(force=True)`` checks that
        alpha is bit-identical to the one the prefetched arena was merged with (else it merges again, synchronously), waits for the
        side stream's event and swaps the arenas -- no kernel of its own on the critical path.  Every step still gets its own merge,
        executed inside the step before; nothing is cached across steps.  +1 arena (+ its bf16 / fp16 pieces) of HBM."""
        if not on:
            self._pipe = None
            return self
        if not hasattr(self.model, "bind_arena"):
            raise RuntimeError("pipelined merges need a HIP encoder model (bind_arena)")
        other = torch.zeros_like(self._arena)
        self._pipe = dict(stream=torch.cuda.Stream(device=self._arena.device), spare=other, spare_ws=None, pending=None)
        return self

    def _alpha_key(self):
        """(storage address, in-place version) of every coefficient tensor: changes whenever an optimizer step, ``load_weights_from_dict``
        or an assignment touches alpha (``load_weights_from_dict`` also drops a pending prefetch outright)."""
        return tuple((p.data_ptr(), p._version) for table in (self.global_weights, self.global_biases, self.per_weights) for p in table.values())

    def _swap_arena(self, arena: torch.Tensor, weightset):
        """make ``arena`` the one the model reads (pointer swaps only); returns the previous (arena, WeightSet)"""
        old = (self._arena, self.model._weights)
        self._arena = arena
        self._merged = arena[: self.layout.padded_numel]
        self.model._flat = self._merged
        self.model._weights = weightset
        self.model._views = weightset.views
        return old

    def _pipelined_load(self, alpha: torch.Tensor):
        from ..engine import WeightSet

        pipe, main = self._pipe, torch.cuda.current_stream(self._arena.device)
        pend = pipe["pending"]
        key = self._alpha_key()   # host-side identity of the coefficient tensors (no device read on the step's path)
        hit = pend is not None and pend["key"] == key
        if hit:
            main.wait_event(pend["done"])                     # the prefetched merge + split of THIS step
            pipe["spare"], pipe["spare_ws"] = self._swap_arena(pend["arena"], pend["ws"])
        else:                                                  # first call, or alpha moved: merge now, into the live arena
            if pend is not None:                               # the speculative arena goes back to being the spare (its merge is ordered
                pipe["spare"], pipe["spare_ws"] = pend["arena"], pend["ws"]  # before the next one on the same side stream)
                pipe["pending"] = None
            self._merge_task_vectors()
            self.model.weights_updated()
        # the next step's merge, under this step's encoder kernels: the spare arena was the live one until the swap above, so the side
        # stream first waits for everything enqueued on the main stream so far (the previous step's readers)
        spare = pipe["spare"]
        ws = pipe["spare_ws"]
        if ws is None or ws.flat.data_ptr() != spare.data_ptr():
            ws = WeightSet(self.layout, spare[: self.layout.padded_numel], self.model._weights.mode)
        fence = torch.cuda.Event()
        fence.record(main)
        side = pipe["stream"]
        with torch.cuda.stream(side):
            side.wait_event(fence)
            self._merge_task_vectors(out=spare if self.slice_plan is not None else spare[: self.layout.padded_numel])
            ws.refresh()
            done = torch.cuda.Event()
            done.record(side)
        pipe["pending"] = dict(key=key, arena=spare, ws=ws, done=done)
        pipe["spare"], pipe["spare_ws"] = None, None

    def load_weights(self, force: bool = False):
        """Re-merge into the model's arena (``force``: unconditionally, as the reference does on every forward) -- skipped when alpha is bit-identical to the one the arena was merged with AND nothing
        else has written the arena since (the model counts in-place writes: load_state_dict, optimizer steps).  The reference
        re-merges on every forward; a catalog encode is hundreds of forwards with the same alpha."""
        alpha = self.effective_alpha().detach()
        if force and getattr(self, "_pipe", None) is not None:
            self._pipelined_load(alpha)
            self._merged_alpha = alpha.clone()
            self.model.arena_changed()
            self._merged_version = getattr(self.model, "_arena_version", 0)
            return self.model
        cached = getattr(self, "_merged_alpha", None)
        version = getattr(self.model, "_arena_version", 0)
        if not force and cached is not None and cached.shape == alpha.shape and getattr(self, "_merged_version", None) == version \
                and torch.equal(cached, alpha):
            return self.model
        self._merge_task_vectors()
        self._merged_alpha = alpha.clone()
        if hasattr(self.model, "weights_updated"):
            self.model.weights_updated()
        self._merged_version = getattr(self.model, "_arena_version", 0)
        return self.model

    def get_state_dict(self) -> StateDict:
        """Named views of a freshly merged flat vector (utils.py:29-40); independent of later merges."""
        merged = self._merge_task_vectors(out=torch.empty_like(self._arena))
        return self.layout.views(merged)

    # -- helpers ---------------------------------------------------------------------------------
    def compact_base(self) -> torch.Tensor:
        self._need_whole("compact_base")
        return self.layout.compact(self.base_model_tensor.data)

    def compact_task_vectors(self) -> torch.Tensor:
        self._need_whole("compact_task_vectors")
        return torch.stack([self.layout.compact(t) for t in self.task_vectors_tensor.data])

    def _need_whole(self, what: str):
        if self.slice_plan is not None:
            raise RuntimeError(f"{what}() needs the whole vectors; this module holds arena slice {self._slice} (placement='sliced')")


class TaskVectorMergingModuleTaskWise(TaskVectorMergingModuleBase):
    def __init__(self, base_model_tensor, task_vectors_tensor, model_without_params, layout, initial_global_weight=1.0,
                 initial_global_bias=0.0, initial_per_weight=0.2, disable_softmax=True, slice_plan=None):
        super().__init__(base_model_tensor, task_vectors_tensor, model_without_params, layout, disable_softmax, slice_plan)
        dev = base_model_tensor.device
        n = task_vectors_tensor.size(0)
        self.global_weights["all"] = nn.Parameter(torch.full((1,), float(initial_global_weight), device=dev))
        self.global_biases["all"] = nn.Parameter(torch.full((1,), float(initial_global_bias), device=dev))
        self.per_weights["all"] = nn.Parameter(torch.full((n,), float(initial_per_weight), device=dev))


class TaskVectorMergingModuleLayerWise(TaskVectorMergingModuleBase):
    def __init__(self, base_model_tensor, task_vectors_tensor, model_without_params, layout, initial_global_weight=1.0,
                 initial_global_bias=0.0, initial_per_weight=0.2, disable_softmax=False, slice_plan=None):
        super().__init__(base_model_tensor, task_vectors_tensor, model_without_params, layout, disable_softmax, slice_plan)
        dev = base_model_tensor.device
        n = task_vectors_tensor.size(0)
        groups, seg_off, seg_gid = layout.group_segments()
        if slice_plan is not None:
            seg_off[-1] = slice_plan.padded  # the last rank's slice runs past the arena into zero padding
        self._groups, self._seg_gid = groups, seg_gid
        self._seg_off = seg_off.to(dev) if len(seg_gid) > 1 else None
        self.layer_groups = groups
        for key in groups:
            self.global_weights[key] = nn.Parameter(torch.full((1,), float(initial_global_weight), device=dev))
            self.global_biases[key] = nn.Parameter(torch.full((1,), float(initial_global_bias), device=dev))
            self.per_weights[key] = nn.Parameter(torch.full((n,), float(initial_per_weight), device=dev))


def load_merging_module(
    merge_type: MergeType,
    learn_type: LearnType,
    model: torch.nn.Module,
    pretrain_state_dict: StateDict,
    finetune_state_dicts: List[StateDict],
    ignore_keys: set,
    ties_density: Optional[float] = None,
    initial_global_weight: float = 1.0,
    initial_global_bias: float = 0.0,
    initial_per_weight: float = 0.2,
    disable_softmax: bool = False,
    device: Optional[torch.device] = None,
    placement: Optional[str] = None,
) -> TaskVectorMergingModuleBase:
    """_factory.py:27-127.  Key order = the pretrained dict's insertion order restricted to keys also in
    ``finetune_state_dicts[0]``.  Like the reference (make_functional, :70) this takes the passed model
    over: afterwards the model computes with the merged arena owned by the returned module.

    ``placement`` (new; the reference is single-GPU): with ``torch.distributed`` initialised on several ranks, "sliced" (the default
    there; MERGEREC_MERGE_PLACEMENT overrides) keeps only this rank's 1/world slice of the base vector and of every task vector on
    the device -- ``get_state_dict()`` / ``load_weights()`` merge that slice and all-gather the arena; "replicated" keeps everything
    on every rank (what alpha learning needs).  One rank: always "replicated"."""
    assert isinstance(merge_type, MergeType), f"Invalid merge type: {merge_type}"
    assert isinstance(learn_type, LearnType), f"Invalid learn type: {learn_type}"
    _check_isinstance_state_dict(pretrain_state_dict)
    for ckpt in finetune_state_dicts:
        _check_isinstance_state_dict(ckpt)

    keys_to_keep = set(pretrain_state_dict.keys() & finetune_state_dicts[0].keys()) - set(ignore_keys)
    pre = OrderedDict((k, v) for k, v in pretrain_state_dict.items() if k in keys_to_keep)
    for ckpt in finetune_state_dicts:  # check_model_shape (model_operations.py:15-44)
        kept = {k for k in ckpt if k in keys_to_keep}
        assert kept == set(pre.keys()), "Models have different architectures."
        for k in pre:
            assert ckpt[k].shape == pre[k].shape, "Models have different shapes."

    if device is None:
        device = getattr(model, "device", None) or torch.device("cuda", torch.cuda.current_device())
    layout = ArenaLayout(OrderedDict((k, tuple(v.shape)) for k, v in pre.items()))

    rank, world_size = parallel.world()
    placement = placement or os.environ.get("MERGEREC_MERGE_PLACEMENT") or ("sliced" if world_size > 1 else "replicated")
    if placement not in ("sliced", "replicated"):
        raise ValueError(f"placement must be 'sliced' or 'replicated', got {placement!r}")
    plan = parallel.SlicePlan(layout.padded_numel, world_size) if (placement == "sliced" and world_size > 1) else None

    print("Converting model to functional form...")
    print("Calculating task vectors...")
    base = layout.pack(pre, device)
    n = len(finetune_state_dicts)
    stage = torch.empty(layout.padded_numel, dtype=torch.float32, device=device)
    direct_slices = plan is not None and merge_type is MergeType.TASK_VECTOR
    if direct_slices:
        # plain task vectors are elementwise: only this rank's slice [lo, hi) of each one is ever formed
        lo, hi = plan.bounds(rank)
        hi_c = min(hi, layout.padded_numel)
        base_s = torch.zeros(plan.slice_len, dtype=torch.float32, device=device)
        base_s[: hi_c - lo] = base[lo:hi_c]
        tv = torch.zeros(n, plan.slice_len, dtype=torch.float32, device=device)
        for i, ckpt in enumerate(finetune_state_dicts):
            layout.pack(ckpt, device, out=stage)
            if hi_c > lo:
                ops.task_vector(stage[lo:hi_c], base[lo:hi_c], out=tv[i, : hi_c - lo])
        base = base_s
    else:
        tv = torch.empty(n, layout.padded_numel, dtype=torch.float32, device=device)
        for i, ckpt in enumerate(finetune_state_dicts):
            layout.pack(ckpt, device, out=stage)
            ops.task_vector(stage, base, out=tv[i])  # algorithms/task_vector.py:8-10
    del stage
    if merge_type is MergeType.TASK_VECTOR:
        pass
    elif merge_type in (MergeType.TIES, MergeType.LOCALIZE_AND_STITCH):
        if merge_type is MergeType.TIES:
            assert ties_density is not None, "Density should be provided for ties merging."
        density = ties_density if ties_density is not None else 0.05  # L&S default (localize_and_stitch.py:9)
        # k counts the reference's flat length (no arena pads); pads are zeros and can only tie with true zeros,
        # whose selection never changes a result (ties.py:16, localize_and_stitch.py:33)
        k = int(density * layout.numel)
        if merge_type is MergeType.TIES:
            for i in range(n):
                ops.abs_topk_mask(tv[i], k, out=tv[i])  # ties.py:15-25
            ops.ties_combine(tv)                           # ties.py:31-72
        else:
            if k <= 0:
                tv.zero_()
            else:
                masks = torch.empty(n, layout.padded_numel, dtype=torch.uint8, device=device)
                scratch = torch.empty(layout.padded_numel, dtype=torch.float32, device=device)
                for i in range(n):
                    _, m = ops.abs_topk_mask(tv[i], k, out=scratch, want_mask=True)
                    masks[i].copy_(m)
                ops.lns_combine(tv, masks, out=tv)          # localize_and_stitch.py:43-49
                del masks, scratch
    elif merge_type is MergeType.PCB:
        # pcb.py:37-58 works on order statistics of each row, so it runs on the COMPACT vectors (arena pads would shift the
        # quantiles); the result is scattered back into arena layout.  density defaults to 0.2 (pcb.py:37).
        density = ties_density if ties_density is not None else 0.2
        compact = torch.stack([layout.compact(tv[i]) for i in range(n)]).contiguous()
        pcb = ops.pcb_vectors(compact, density)
        tv.zero_()
        off = 0
        for k, shp in layout.shapes.items():
            cnt = 1
            for x_ in shp:
                cnt *= x_
            tv[:, layout.offsets[k] : layout.offsets[k] + cnt] = pcb[:, off : off + cnt]
            off += cnt
        del compact, pcb
    else:
        raise ValueError(f"Invalid merge type: {merge_type}")

    if plan is not None and not direct_slices:
        # TIES / L&S / PCB select over whole vectors (global order statistics): computed on every rank at init, then cut to the slice
        lo, hi = plan.bounds(rank)
        hi_c = min(hi, layout.padded_numel)
        base_s = torch.zeros(plan.slice_len, dtype=torch.float32, device=device)
        tv_s = torch.zeros(n, plan.slice_len, dtype=torch.float32, device=device)
        base_s[: hi_c - lo] = base[lo:hi_c]
        tv_s[:, : hi_c - lo] = tv[:, lo:hi_c]
        base, tv = base_s, tv_s

    print("Creating merging module...")
    if learn_type is LearnType.TASK_WISE:
        cls = TaskVectorMergingModuleTaskWise
    elif learn_type is LearnType.LAYER_WISE:
        cls = TaskVectorMergingModuleLayerWise
    else:
        raise ValueError(f"Invalid learn type: {learn_type}")
    return cls(base, tv, model, layout, initial_global_weight=initial_global_weight, initial_global_bias=initial_global_bias,
               initial_per_weight=initial_per_weight, disable_softmax=disable_softmax, slice_plan=plan)
