"""Device-side engine for the merged-model inference path: the parameter arena and the encoder runner.

Data layout in HBM
  * ParamArena: ONE flat fp32 buffer per model holding every tensor of the reference's state_dict in
    the reference's key order (merger/weight_learning/module/_factory.py:55-66), each tensor starting
    on a 64-float (256 B) boundary so every weight/bias/table row is 16-byte aligned for the kernels
    (Recformer's 4098-element ``position_ids`` buffer would otherwise shift everything by 8 bytes).
    Base model, the N task vectors ((N, P_pad) row-major) and the merged model share the layout, so
    the merge is one streaming pass and the encoder reads the merged weights in place (no
    ``load_state_dict`` copy as in merge_test.py:80).
  * Activations: packed tokens, (T, d) fp32 row-major, T = number of non-masked tokens in the batch
    (padding never reaches the GPU kernels), ``cu_seqlens`` int32 (B+1).

The runner issues the HIP kernels of include/mergerec_hip.h on the current stream; it holds no
compute of its own.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import ops

ARENA_ALIGN = 64  # floats


@dataclass
class EncoderSpec:
    """Architecture of one encoder family (dims of BLaIR = RoBERTa, Recformer = Longformer)."""

    kind: str = "roberta"  # "roberta" | "recformer"
    hidden: int = 768
    heads: int = 12
    layers: int = 12
    intermediate: int = 3072
    vocab: int = 50265
    max_pos: int = 514
    pad_id: int = 1
    ln_eps: float = 1e-5
    token_type_size: int = 1
    max_item_embeddings: int = 0
    one_sided_window: int = -1  # recformer: attention_window 64 -> 32 (interface.py:23)
    pooler: bool = True  # RobertaModel carries an (unused) pooler

    @staticmethod
    def blair_base():
        return EncoderSpec()

    @staticmethod
    def blair_large():
        return EncoderSpec(hidden=1024, heads=16, layers=24, intermediate=4096)

    @staticmethod
    def recformer_base():
        return EncoderSpec(kind="recformer", max_pos=4098, token_type_size=4, max_item_embeddings=51, one_sided_window=32, pooler=False)

    @staticmethod
    def recformer_large():
        return EncoderSpec(kind="recformer", hidden=1024, heads=16, layers=24, intermediate=4096, max_pos=4098,
                           token_type_size=4, max_item_embeddings=51, one_sided_window=32, pooler=False)

    def param_shapes(self, prefix: str = "model.") -> "OrderedDict[str, Tuple[int, ...]]":
        """state_dict key order of the reference wrapper (transformers 4.51.3 module order)."""
        d, i = self.hidden, self.intermediate
        sh: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
        e = prefix + "embeddings."
        if self.kind == "recformer":
            sh[e + "position_ids"] = (1, self.max_pos)  # persistent int64 buffer (recformer/models.py:96)
        sh[e + "word_embeddings.weight"] = (self.vocab, d)
        sh[e + "position_embeddings.weight"] = (self.max_pos, d)
        sh[e + "token_type_embeddings.weight"] = (self.token_type_size, d)
        if self.kind == "recformer":
            sh[e + "item_position_embeddings.weight"] = (self.max_item_embeddings, d)
        sh[e + "LayerNorm.weight"] = (d,)
        sh[e + "LayerNorm.bias"] = (d,)
        projs = ("query", "key", "value") + (("query_global", "key_global", "value_global") if self.kind == "recformer" else ())
        for l in range(self.layers):
            lp = f"{prefix}encoder.layer.{l}."
            for n in projs:
                sh[lp + f"attention.self.{n}.weight"] = (d, d)
                sh[lp + f"attention.self.{n}.bias"] = (d,)
            sh[lp + "attention.output.dense.weight"] = (d, d)
            sh[lp + "attention.output.dense.bias"] = (d,)
            sh[lp + "attention.output.LayerNorm.weight"] = (d,)
            sh[lp + "attention.output.LayerNorm.bias"] = (d,)
            sh[lp + "intermediate.dense.weight"] = (i, d)
            sh[lp + "intermediate.dense.bias"] = (i,)
            sh[lp + "output.dense.weight"] = (d, i)
            sh[lp + "output.dense.bias"] = (d,)
            sh[lp + "output.LayerNorm.weight"] = (d,)
            sh[lp + "output.LayerNorm.bias"] = (d,)
        if self.kind == "roberta" and self.pooler:
            sh[prefix + "pooler.dense.weight"] = (d, d)
            sh[prefix + "pooler.dense.bias"] = (d,)
        return sh


class ArenaLayout:
    """name -> (offset, numel, shape) for one key order; offsets are multiples of 64 floats."""

    def __init__(self, shapes: "OrderedDict[str, Sequence[int]]"):
        self.shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict((k, tuple(int(x) for x in v)) for k, v in shapes.items())
        self.offsets: Dict[str, int] = {}
        off = 0
        for k, shp in self.shapes.items():
            self.offsets[k] = off
            n = math.prod(shp)
            off += (n + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
        self.padded_numel = off
        self.numel = sum(math.prod(s) for s in self.shapes.values())  # algorithmic P (reference's flat length)

    def keys(self):
        return list(self.shapes.keys())

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        o, shp = self.offsets[name], self.shapes[name]
        return flat[o : o + math.prod(shp)].view(shp)

    def views(self, flat: torch.Tensor) -> "OrderedDict[str, torch.Tensor]":
        """a6: named views of a flat arena vector (merger/weight_learning/utils.py:29-40), zero-copy."""
        return OrderedDict((k, self.view(flat, k)) for k in self.shapes)

    def pack(self, state_dict: Dict[str, torch.Tensor], device, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """a2: flatten a state_dict into arena layout (model_operations.py:47-63 semantics: key order of
        this layout, int buffers promoted to fp32).  Host packs into one pinned staging buffer, one H2D."""
        missing = [k for k in self.shapes if k not in state_dict]
        if missing:
            raise KeyError(f"state_dict is missing keys: {missing[:3]}{'...' if len(missing) > 3 else ''}")
        all_dev = all(state_dict[k].is_cuda for k in self.shapes)
        if out is None:
            out = torch.zeros(self.padded_numel, dtype=torch.float32, device=device)
        if all_dev:
            for k, shp in self.shapes.items():
                v = state_dict[k]
                if tuple(v.shape) != shp:
                    raise ValueError(f"shape mismatch for {k}: {tuple(v.shape)} vs {shp}")
                self.view(out, k).copy_(v)
            return out
        stage = torch.zeros(self.padded_numel, dtype=torch.float32)
        for k, shp in self.shapes.items():
            v = state_dict[k]
            if tuple(v.shape) != shp:
                raise ValueError(f"shape mismatch for {k}: {tuple(v.shape)} vs {shp}")
            o = self.offsets[k]
            stage[o : o + v.numel()].copy_(v.detach().reshape(-1))
        out.copy_(stage, non_blocking=False)
        return out

    def compact(self, flat: torch.Tensor) -> torch.Tensor:
        """The reference's contiguous (P,) flat vector (no pads), for API compatibility."""
        return torch.cat([self.view(flat, k).reshape(-1) for k in self.shapes])

    def group_segments(self) -> Tuple[List[str], torch.Tensor, List[int]]:
        """a5: layer-wise groups (layer_wise.py:13-33: ``name.split('.')[3]`` if 'encoder.layer.' in name
        else 'others'), as maximal runs of adjacent same-group tensors in arena coordinates.
        Returns (group keys in first-seen order, seg_off int64 (S+1), group index per segment)."""
        memo = getattr(self, "_group_segments", None)
        if memo is not None:
            return memo[0], memo[1].clone(), memo[2]
        groups: List[str] = []
        seg_start: List[int] = []
        seg_gid: List[int] = []
        for k in self.shapes:
            key = k.split(".")[3] if "encoder.layer." in k else "others"
            if key not in groups:
                groups.append(key)
            gid = groups.index(key)
            if not seg_gid or seg_gid[-1] != gid:
                seg_start.append(self.offsets[k])
                seg_gid.append(gid)
        seg_off = torch.tensor(seg_start + [self.padded_numel], dtype=torch.int64)
        self._group_segments = (groups, seg_off.clone(), seg_gid)
        return groups, seg_off, seg_gid


GEMM_MODES = ("f32", "bf16x6", "f16x3", "bf16x3")
_PRODUCTS = {"f32": 0, "bf16x6": 6, "bf16x3": 3, "f16x3": ops.PRODUCTS_F16X3}


def _attn_products(mode: str) -> int:
    """products of the attention kernels for an encoder arithmetic (0 = exact fp32).  MERGEREC_ATTN_PRODUCTS overrides it for error
    attribution runs (tests/tools/trained_like_attribution.py: which half of an arithmetic's distance is the linears', which attention's)."""
    import os

    o = os.environ.get("MERGEREC_ATTN_PRODUCTS")
    return int(o) if o in ("0", "3", "6", "35") else _PRODUCTS[mode]


def default_gemm_mode() -> str:
    """Encoder GEMM arithmetic: "bf16x6" = six bf16 MFMA products per fp32 product (fp32-grade accuracy, 2.67x fewer
    matrix-pipe cycles; the library default: no range limits), "f16x3" = three products of two FP16 pieces per operand (~2^-21 per
    product at bf16x3's cost; fp32-grade on trained-like weights, fixture g22; operands must stay inside fp16's range -- activations
    |x| < 65504, weights |w| < 255.9, checked), "bf16x3" = three products of two bf16 pieces (~2^-16 per product: inside 1e-4 on
    init-like weights, NOT on trained-like ones -- opt-in only), "f32" = exact fp32 MFMA (bit-identical to the oracle's FMA chain)."""
    import os

    m = os.environ.get("MERGEREC_GEMM_MODE", "bf16x6")
    if m not in GEMM_MODES:
        raise ValueError(f"MERGEREC_GEMM_MODE must be one of {GEMM_MODES}")
    return m


class WeightSet:
    """What the encoder reads: named fp32 views of one arena (+ for "bf16x6" the three bf16 piece arenas that
    mirror it element for element).  refresh() must be called after the arena's contents change (merge, load)."""

    def __init__(self, layout: "ArenaLayout", flat: torch.Tensor, mode: Optional[str] = None):
        self.layout, self.flat = layout, flat
        self.views = layout.views(flat)
        self.mode = default_gemm_mode() if mode is None else mode
        if self.mode not in GEMM_MODES:
            raise ValueError(f"gemm mode must be one of {GEMM_MODES}")
        self.pieces = None
        self._table = None
        self._overflow = None  # f16x3: device flag raised by the weight split when a scaled weight leaves fp16's range

    def __getitem__(self, k):
        return self.views[k]

    def __contains__(self, k):
        return k in self.views

    def offset(self, name: str) -> int:
        return self.layout.offsets[name]

    def refresh(self):
        if self.mode in ("bf16x6", "bf16x3", "f16x3"):
            if self._table is None:  # every 2-D tensor a Linear reads (not the embedding tables), K % 16 == 0
                ent = [(self.layout.offsets[k], shp[0], shp[1]) for k, shp in self.layout.shapes.items()
                       if len(shp) == 2 and "embeddings" not in k and shp[1] % 16 == 0]
                self._table = ops.KBlockTable(ent, self.flat.device)
            if self.mode == "f16x3":
                if self._overflow is None:
                    self._overflow = torch.zeros(1, dtype=torch.int32, device=self.flat.device)
                self.pieces = ops.split_weights_kblock(self.flat, self._table, self.pieces, f16=True, overflow=self._overflow)
            else:
                self.pieces = ops.split_weights_kblock(self.flat, self._table, self.pieces, n_pieces=3 if self.mode == "bf16x6" else 2)
        return self

    def check_range(self):
        """f16x3 only: raise if a weight split since the last call met a value outside fp16's range (one device read; the evaluation
        loops call it where they synchronise anyway, through ``EncoderRunner.check_inputs``)."""
        if self._overflow is not None and int(self._overflow.item()):
            self._overflow.zero_()
            raise InputError("a weight matrix holds |w| >= 255.9 (or NaN): outside the fp16 pieces of the 'f16x3' arithmetic -- "
                             "use gemm_mode='bf16x6' (no range limit) for this model")

    @staticmethod
    def from_views(views: Dict[str, torch.Tensor]) -> "WeightSet":
        """Plain dict of tensors (tests): packs them into a fresh arena."""
        layout = ArenaLayout(OrderedDict((k, tuple(v.shape)) for k, v in views.items()))
        dev = next(iter(views.values())).device
        flat = layout.pack(views, dev)
        return WeightSet(layout, flat).refresh()


# ------------------------------------------------------------------------------------------------
@dataclass
class PackedBatch:
    """A batch after host-side length analysis; everything the kernels need."""

    B: int
    T: int
    max_len: int
    cu_seqlens: torch.Tensor  # int32 (B+1) device
    cls_rows: torch.Tensor  # int32 (B) device == cu_seqlens[:-1]
    tok_word: torch.Tensor
    tok_pos: torch.Tensor
    tok_tt: Optional[torch.Tensor] = None
    tok_ip: Optional[torch.Tensor] = None
    sum_len_sq: float = 0.0  # sum_b L_b^2 (algorithmic attention work, for the profiler)
    seq_order: Optional[torch.Tensor] = None  # int32 (B): sequence ids by decreasing length (attention scheduling hint)
    attn_work: Optional[Dict[int, Tuple[torch.Tensor, int]]] = None  # q_rows -> (device int32 work list, n_slots): ops.attn_work_plan
    pad_len: Optional[torch.Tensor] = None  # int32 (B) device: padded width of each row's source batch (pooling_method="mean" only)


def _lens_from_mask(attention_mask: torch.Tensor) -> torch.Tensor:
    # one small D2H sync when the mask lives on the GPU; free when the collator's CPU tensors are passed
    return attention_mask.ne(0).sum(dim=1).to("cpu", torch.int64)


class InputError(ValueError):
    """A batch violated the encoder's input contract (found by the packing kernel, reported at the next ``check_inputs``)."""


def check_module_inputs(*objs) -> None:
    """Surface deferred input-contract violations (``InputError``) of every encoder reachable from the given modules: the object itself,
    its ``.model`` / ``.merged_model`` and theirs.  ONE device read per encoder: the loops call it where they synchronise anyway (after a
    catalog encode, at a logged training step, at an epoch end)."""
    seen, todo = set(), [o for o in objs if o is not None]
    while todo:
        o = todo.pop()
        if id(o) in seen:
            continue
        seen.add(id(o))
        runner = getattr(o, "runner", None)
        if isinstance(runner, EncoderRunner):
            runner.check_inputs()
        for name in ("model", "merged_model"):
            child = getattr(o, name, None)
            if child is not None and not isinstance(child, (str, bytes)):
                todo.append(child)


class EncoderRunner:
    """Runs the BLaIR (RoBERTa) or Recformer (Longformer) forward on packed tokens with HIP kernels.

    ``weights`` is any mapping name -> device tensor (normally ArenaLayout.views(merged_flat))."""

    def __init__(self, spec: EncoderSpec, prefix: str = "model."):
        if spec.hidden % spec.heads or spec.hidden // spec.heads != 64:
            raise ValueError("attention kernels are built for head_dim == 64 (BLaIR/Recformer base and large)")
        self.spec = spec
        self.prefix = prefix
        self.fuse_qkv = spec.hidden % 128 == 0
        self._err_bits: Dict[torch.device, torch.Tensor] = {}  # one int32 word per device, OR-ed into by the packing kernel

    # ---- batch preparation -----------------------------------------------------------------------
    def _err_word(self, device) -> torch.Tensor:
        device = torch.device(device)
        w = self._err_bits.get(device)
        if w is None:
            w = self._err_bits[device] = torch.zeros(1, dtype=torch.int32, device=device)
        return w

    def check_inputs(self) -> None:
        """Raise InputError if any batch packed since the last call broke the input contract.  ONE device -> host read; the callers
        place it where they synchronise anyway (end of a catalog encode, end of an evaluation epoch)."""
        for w in self._err_bits.values():
            bits = int(w.item())
            if bits:
                w.zero_()
                raise InputError("; ".join(msg for bit, msg in ops.INPUT_ERRORS.items() if bits & bit))

    def pack(self, batch: Dict[str, torch.Tensor], device, lens: Optional[torch.Tensor] = None, validate=True) -> PackedBatch:
        """validate: True = range / pattern checks run inside the packing kernel and surface at the next ``check_inputs()`` (no host
        sync here); "now" = the same, checked immediately (one sync); False = unchecked."""
        ids, mask = batch["input_ids"], batch["attention_mask"]
        if ids.dim() != 2 or ids.shape != mask.shape:
            raise ValueError("input_ids / attention_mask must be (B, L) and equal-shaped")
        B, L = ids.shape
        if lens is None:
            lens = getattr(batch, "host_lens", None)  # stashed by ToDeviceMixin.to() while the mask was still on the host
        if lens is None:
            lens = _lens_from_mask(mask)
        lens = lens.to(torch.int64).cpu()
        if lens.numel() != B:
            raise ValueError("lens must hold one length per row")
        if validate and B > 0:
            if int(lens.min()) < 1:
                raise ValueError("every sequence needs at least one attended token (CLS)")
            if L + self.spec.pad_id + 1 > self.spec.max_pos:
                raise ValueError(f"sequence length {L} exceeds the position table ({self.spec.max_pos})")
        cu = torch.zeros(B + 1, dtype=torch.int32)
        if B:
            cu[1:] = lens.cumsum(0).to(torch.int32)
        T = int(cu[-1])
        cu_d = ops.h2d(cu, device)
        tt = ip = gm = None
        if self.spec.kind == "recformer":
            for key in ("token_type_ids", "item_position_ids", "global_attention_mask"):
                if key not in batch:
                    raise ValueError(f"Missing required key in batch: {key}")  # interface.py:71-74
            tt, ip, gm = batch["token_type_ids"], batch["item_position_ids"], batch["global_attention_mask"]
        dev = lambda t: None if t is None else (t.to(device, torch.int64) if t.is_cuda else ops.h2d(t.to(torch.int64), device)).contiguous()
        err = self._err_word(device) if validate else None
        tw, tp, ttp, tip = ops.pack_tokens(dev(ids), dev(mask), cu_d, T, self.spec.pad_id, dev(tt), dev(ip), dev(gm) if validate else None, err,
                                           self.spec.vocab, self.spec.token_type_size, self.spec.max_item_embeddings)
        if validate == "now":
            self.check_inputs()
        # work lists of the split attention kernels (host-built from the lengths, ONE small copy for both block heights)
        attn_work = None
        if B:
            plans = [ops.attn_work_plan(lens, q) for q in (128, 256)]
            both = ops.h2d(torch.cat([p[0] for p in plans]), device)
            n128 = plans[0][0].numel()
            attn_work = {128: (both[:n128], plans[0][1]), 256: (both[n128:], plans[1][1])}
        pad_len = None
        if getattr(self, "pooling_method", "cls") == "mean":
            hp = getattr(batch, "host_pad_len", None)
            hp = torch.full((B,), L, dtype=torch.int64) if hp is None else torch.as_tensor(hp).to(torch.int64).cpu()
            if hp.numel() != B or (B and bool((hp < lens).any())):
                raise ValueError("host_pad_len must hold one padded width per row, none shorter than its row's length")
            pad_len = ops.h2d(hp.to(torch.int32), device)
        return PackedBatch(B=B, T=T, max_len=int(lens.max()) if B else 0, cu_seqlens=cu_d, cls_rows=cu_d[:-1].contiguous(), attn_work=attn_work, pad_len=pad_len,
                           tok_word=tw, tok_pos=tp, tok_tt=ttp, tok_ip=tip, sum_len_sq=float((lens.double() ** 2).sum()) if B else 0.0,
                           seq_order=ops.h2d(torch.argsort(lens, descending=True, stable=True).to(torch.int32), device) if B > 1 else None)

    # ---- forward ---------------------------------------------------------------------------------
    def embed(self, w: Dict[str, torch.Tensor], pb: PackedBatch) -> torch.Tensor:
        e = self.prefix + "embeddings."
        rec = self.spec.kind == "recformer"
        return ops.embed_gather_ln(
            pb.tok_word, pb.tok_pos, pb.tok_tt, pb.tok_ip, w[e + "word_embeddings.weight"], w[e + "position_embeddings.weight"],
            w[e + "token_type_embeddings.weight"], w[e + "item_position_embeddings.weight"] if rec else None,
            w[e + "LayerNorm.weight"], w[e + "LayerNorm.bias"], self.spec.ln_eps, ops.EMBED_RECFORMER if rec else ops.EMBED_ROBERTA)

    @staticmethod
    def _ws(w) -> "WeightSet":
        return w if isinstance(w, WeightSet) else WeightSet.from_views(dict(w))

    def _linear(self, w: "WeightSet", x, wnames, bnames, act=ops.ACT_NONE, residual=None, out=None):
        """act(x @ W_s^T + b_s) (+ residual) for 1..3 equally shaped weight segments, in the WeightSet's GEMM mode."""
        biases = [w[b] for b in bnames]
        seg_n, K = w[wnames[0]].shape
        fused = len(wnames) == 1 or seg_n % 128 == 0
        if not fused:
            out = torch.empty(x.shape[0], len(wnames) * seg_n, dtype=torch.float32, device=x.device) if out is None else out
            for i, (wn_, bn_) in enumerate(zip(wnames, bnames)):
                self._linear(w, x, [wn_], [bn_], act, None, out[:, i * seg_n : (i + 1) * seg_n])
            return out
        if w.mode in ("bf16x6", "bf16x3", "f16x3"):
            if w.pieces is None:
                w.refresh()
            return ops.gemm_nt_split(x, w.pieces, [w.offset(n) for n in wnames], seg_n, K, biases, act, residual, out,
                                     products=_PRODUCTS[w.mode])
        return ops.gemm_nt(x, [w[n] for n in wnames], biases, act, residual, out)

    def _proj(self, w, lp, names, x):
        return self._linear(w, x, [f"{lp}attention.self.{n}.weight" for n in names], [f"{lp}attention.self.{n}.bias" for n in names])

    def layer(self, w: Dict[str, torch.Tensor], l: int, x: torch.Tensor, pb: PackedBatch, cls_only: bool = False) -> torch.Tensor:
        """One post-LN transformer block.  cls_only: after attention keep only each sequence's first row
        (a13: the last layer's output is read at [:, 0] only, encoder/_base.py:45)."""
        sp, lp = self.spec, f"{self.prefix}encoder.layer.{l}."
        rec = sp.kind == "recformer"
        if cls_only:
            # a13: only the first row of every sequence is read after this layer, so attention runs for those B queries alone: the
            # query projection on the CLS rows, key / value projections on all tokens, one exact-fp32 attention row per (sequence, head).
            # RoBERTa uses its query / key / value weights; Longformer's CLS row is the global row (*_global weights) and its windowed
            # rows are not needed at all.
            names = ("query_global", "key_global", "value_global") if rec else ("query", "key", "value")
            x_cls = ops.gather_rows(x, pb.cls_rows)
            q = self._proj(w, lp, names[:1], x_cls)
            kv = self._proj(w, lp, names[1:], x)
            ctx = torch.empty(pb.B, sp.hidden, dtype=torch.float32, device=x.device)
            ops.attention_global_row(q, kv, pb.cu_seqlens, pb.B, sp.heads, pb.max_len, ctx, compact=True)
            x = x_cls
        else:
            qkv = self._proj(w, lp, ("query", "key", "value"), x)
            if ops.PROF.enabled:
                w_ = sp.one_sided_window
                ops.ATTN_FLOPS_HINT[0] = 4.0 * sp.hidden * (pb.sum_len_sq if not rec else pb.T * (2 * w_ + 2))
            chain = ops.PROF.chain if (self.fuse_qkv and not rec) else (lambda: None)  # back-to-back launches: shared event stamps
            chain()
            ctx = ops.attention(qkv, pb.cu_seqlens, pb.B, sp.heads, pb.max_len, window=sp.one_sided_window if rec else -1,
                                seq_order=pb.seq_order, products=_attn_products(w.mode), work=pb.attn_work)
            if rec:
                qg = self._proj(w, lp, ("query_global",), ops.gather_rows(x, pb.cls_rows))
                kvg = self._proj(w, lp, ("key_global", "value_global"), x)
                ops.attention_global_row(qg, kvg, pb.cu_seqlens, pb.B, sp.heads, pb.max_len, ctx)
        chain = ops.PROF.chain if (not cls_only and not rec and self.fuse_qkv) else (lambda: None)
        chain()
        h = self._linear(w, ctx, [lp + "attention.output.dense.weight"], [lp + "attention.output.dense.bias"], residual=x)
        chain()
        h = ops.layernorm(h, w[lp + "attention.output.LayerNorm.weight"], w[lp + "attention.output.LayerNorm.bias"], sp.ln_eps, out=h)
        chain()
        i = self._linear(w, h, [lp + "intermediate.dense.weight"], [lp + "intermediate.dense.bias"], act=ops.ACT_GELU)
        chain()
        o = self._linear(w, i, [lp + "output.dense.weight"], [lp + "output.dense.bias"], residual=h)
        chain()
        return ops.layernorm(o, w[lp + "output.LayerNorm.weight"], w[lp + "output.LayerNorm.bias"], sp.ln_eps, out=o)

    def _pad_row_layer(self, w, l: int, x_pad: torch.Tensor, kv: torch.Tensor, pb: PackedBatch) -> torch.Tensor:
        """One block for the pad positions' hidden state (pooling_method="mean"): every pad position of sequence b holds the same vector
        -- the pad token's embedding, attending to the sequence's valid keys, never a key itself -- so ONE query row per sequence, against
        the layer's keys / values ``kv`` (T, 2 d) of the real tokens, reproduces what upstream computes at all of them."""
        sp, lp = self.spec, f"{self.prefix}encoder.layer.{l}."
        q = self._proj(w, lp, ("query",), x_pad)
        ctx = torch.empty(pb.B, sp.hidden, dtype=torch.float32, device=x_pad.device)
        ops.attention_global_row(q, kv, pb.cu_seqlens, pb.B, sp.heads, pb.max_len, ctx, compact=True)
        h = self._linear(w, ctx, [lp + "attention.output.dense.weight"], [lp + "attention.output.dense.bias"], residual=x_pad)
        h = ops.layernorm(h, w[lp + "attention.output.LayerNorm.weight"], w[lp + "attention.output.LayerNorm.bias"], sp.ln_eps, out=h)
        i = self._linear(w, h, [lp + "intermediate.dense.weight"], [lp + "intermediate.dense.bias"], act=ops.ACT_GELU)
        o = self._linear(w, i, [lp + "output.dense.weight"], [lp + "output.dense.bias"], residual=h)
        return ops.layernorm(o, w[lp + "output.LayerNorm.weight"], w[lp + "output.LayerNorm.bias"], sp.ln_eps, out=o)

    def _forward_mean(self, w, pb: PackedBatch, normalize: bool) -> torch.Tensor:
        """pooling_method="mean" (encoder/_base.py:42-43): every layer on all tokens plus the pad row of each sequence; then
        (sum of the sequence's rows + (padded width - length) x pad row) / padded width."""
        sp = self.spec
        if sp.kind != "roberta":
            raise NotImplementedError("pooling_method='mean' is built for the RoBERTa-family encoders (Longformer pads to window multiples first)")
        if pb.pad_len is None:
            raise RuntimeError("pooling_method='mean' needs the padded widths (EncoderRunner.pack with pooling_method == 'mean')")
        dev = pb.cu_seqlens.device
        x = self.embed(w, pb)
        pad_ids = torch.full((pb.B,), sp.pad_id, dtype=torch.int32, device=dev)  # RobertaEmbeddings: pad tokens sit at position padding_idx
        e = self.prefix + "embeddings."
        x_pad = ops.embed_gather_ln(pad_ids, pad_ids, None, None, w[e + "word_embeddings.weight"], w[e + "position_embeddings.weight"],
                                    w[e + "token_type_embeddings.weight"], None, w[e + "LayerNorm.weight"], w[e + "LayerNorm.bias"], sp.ln_eps,
                                    ops.EMBED_ROBERTA)
        d = sp.hidden
        for l in range(sp.layers):
            lp = f"{self.prefix}encoder.layer.{l}."
            kv = self._proj(w, lp, ("key", "value"), x)  # the real tokens' keys / values of this layer, as the pad rows see them
            x_pad = self._pad_row_layer(w, l, x_pad, kv, pb)
            x = self.layer(w, l, x, pb, cls_only=False)
        return ops.mean_pool(x, pb.cu_seqlens, x_pad, pb.pad_len, pb.B, normalize)

    def forward_packed(self, w, pb: PackedBatch, normalize: bool, return_hidden: bool = False):
        """-> (B, d) CLS embeddings (L2-normalised if ``normalize``); optionally every layer's packed hidden.
        ``w`` is a WeightSet (or a plain name -> tensor mapping, packed on the fly)."""
        w = self._ws(w)
        d = self.spec.hidden
        if pb.B == 0:
            empty = torch.empty(0, d, dtype=torch.float32, device=pb.cu_seqlens.device)
            return (empty, []) if return_hidden else empty
        if getattr(self, "pooling_method", "cls") == "mean" and not return_hidden:
            return self._forward_mean(w, pb, normalize)
        x = self.embed(w, pb)
        hidden = [x] if return_hidden else None
        L = self.spec.layers
        for l in range(L):
            last = l == L - 1
            if l and not (last and not return_hidden):
                ops.PROF.chain()  # a full layer's first launch follows the previous layer's LayerNorm directly
            x = self.layer(w, l, x, pb, cls_only=last and not return_hidden)
            if return_hidden:
                hidden.append(x)
        pooler = getattr(self, "pooling_method", "cls") == "pooler"
        if return_hidden or L == 0:
            out = ops.cls_pool_normalize(x, pb.cu_seqlens, pb.B, normalize and not pooler)
        else:  # x already holds one row per sequence
            out = ops.cls_pool_normalize(x, None, pb.B, normalize and not pooler)
        if pooler:
            # encoder/_base.py:46-47: ``outputs.pooler_output`` = RobertaPooler: tanh(dense(h[:, 0])), exact fp32 (a (B, d) product)
            p = self.prefix + "pooler.dense."
            if p + "weight" not in w:
                raise RuntimeError("pooling_method='pooler' needs a model with a pooler head (Recformer has none: upstream returns None there)")
            out = ops.gemm_nt(out, [w[p + "weight"]], [w[p + "bias"]], act=ops.ACT_TANH)
            if normalize:
                out = ops.cls_pool_normalize(out, None, pb.B, True)
        return (out, hidden) if return_hidden else out

    def encode(self, w: Dict[str, torch.Tensor], batch: Dict[str, torch.Tensor], device, normalize: bool, lens=None, validate: bool = True):
        return self.forward_packed(w, self.pack(batch, device, lens=lens, validate=validate), normalize)


# ------------------------------------------------------------------------------------------------
def flops_per_sequence(spec: EncoderSpec, L: int, cls_only_last: bool = True) -> float:
    """Algorithmic FLOPs of one encoder forward: layers * (24 L d^2 + 4 L^2 d) for RoBERTa."""
    d = spec.hidden
    per_layer = 24 * L * d * d + 4 * L * L * d
    if spec.kind == "recformer":
        per_layer = 28 * L * d * d + 4 * L * 66 * d + 4 * L * d
    return float(spec.layers * per_layer)
