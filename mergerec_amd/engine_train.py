"""Differentiable BLaIR (RoBERTa) / Recformer (Longformer) encoder for the collaborative-merging optimisation loop (merge_train.py, HOT LOOP 3 of
SURVEY.md §3.2): forward on packed tokens with the activations kept, backward producing d loss / d merged parameters as ONE
flat vector in the parameter arena's layout -- exactly the ``g`` that ``mr_merge_bwd_alpha_f32`` contracts with the task vectors.

What the reference does there: ``make_functional`` + torch autograd through transformers' RobertaModel
(merger/weight_learning/_base.py:78-81, module/distiller/sequence/module.py:76-79).  Here every product is the exact-fp32 NT
GEMM of the inference path (gemm.hip) after an operand re-layout (csrc/backward.hip), attention backward is a pair of
query-owned / key-owned kernels, LayerNorm / GELU / bias gradients are row and column kernels.  Batches are tiny (16 pseudo-user
sequences of item text), so the step is bound by the parameter-sized streams, not by token math.

Recformer: the band + global-key attention backward is the same pair of kernels with the Longformer mask; the global CLS row
(query_global / key_global / value_global projections) has its own one-query backward kernel; four embedding tables."""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import ops
from .engine import ArenaLayout, EncoderSpec, PackedBatch

__all__ = ["Dropout", "EncoderTrainGraph", "RobertaTrainGraph", "SplitWeights", "encode_with_grad"]


@dataclass(frozen=True)
class Dropout:
    """Dropout of ONE training forward (and its backward): HF's two rates and the (seed, step) pair that keys the counter-based mask
    (csrc/dropout.h).  The reference's sites: the embedding LayerNorm output (transformers RobertaEmbeddings; recformer/models.py:93,135),
    the attention probabilities, the attention-output and FFN-output dense results before their residual adds (transformers
    RobertaSelfOutput / RobertaOutput, LongformerSelfOutput / LongformerOutput), the Longformer global row's probabilities."""

    p_hidden: float = 0.1
    p_attn: float = 0.1
    seed: int = 0
    step: int = 0

    def key(self, layer: int, site: int) -> int:
        return ops.dropout_site_key(self.seed, self.step, layer, site)

    @property
    def active(self) -> bool:
        return self.p_hidden > 0.0 or self.p_attn > 0.0

# A/B switch of the r04 token-sized tile kernel in the exact-fp32 graph (1 = default; 0 = the r03 route: NT kernel + split-K + transposes)
_TILE = os.environ.get("MR_TRAIN_TILE", "1") != "0"
_TILE_PRODUCTS = int(os.environ.get("MR_TRAIN_TILE_PRODUCTS", "0"))
# weight-gradient products of the tile backward on a second stream (they feed nothing in the backward chain): MR_TRAIN_WGRAD_STREAM=0 keeps one stream
_WGRAD_STREAM = os.environ.get("MR_TRAIN_WGRAD_STREAM", "1") != "0"
# the alpha-learning step's merge and alpha-gradient contraction in arena ranges on that second stream (merger.weight_learning.MergeOverlap)
_MERGE_OVERLAP = os.environ.get("MR_TRAIN_MERGE_OVERLAP", "1") != "0"
# ... with the word-embedding table merged and contracted in the batch's rows only (MR_TRAIN_SPARSE_WORD_ROWS=0: the whole table, as every other range)
_SPARSE_WORD_ROWS = os.environ.get("MR_TRAIN_SPARSE_WORD_ROWS", "1") != "0"

_LINEARS = ("attention.self.query", "attention.self.key", "attention.self.value", "attention.output.dense", "intermediate.dense", "output.dense")


def _pad32(n: int) -> int:
    return (n + 31) // 32 * 32


class SplitWeights:
    """bf16 pieces of the weights for the "bf16x3" training graph (fine-tuning at token-sized batches): the k-blocked split of every
    Linear weight W (forward, y = x W^T) and of its transpose (backward, dx = dy W), re-derived by ``refresh(flat)`` whenever the
    arena changes (once per optimizer step: two split launches + one transpose per weight)."""

    def __init__(self, spec: EncoderSpec, layout: ArenaLayout, prefix: str, device):
        self.spec, self.layout, self.prefix = spec, layout, prefix
        d, di = spec.hidden, spec.intermediate
        rec = spec.kind == "recformer"
        fwd_names = list(_LINEARS) + (["attention.self.query_global", "attention.self.key_global", "attention.self.value_global"] if rec else [])
        ent = []
        for l in range(spec.layers):
            for n in fwd_names:
                k = f"{prefix}encoder.layer.{l}.{n}.weight"
                ent.append((layout.offsets[k],) + tuple(layout.shapes[k]))
        self.table = ops.KBlockTable(ent, device)
        # transposed scratch: per layer [Wq; Wk; Wv]^T (d, 3d) | Wo^T (d, d) | W1^T (d, di) | W2^T (di, d) (| [Wkg; Wvg]^T (d, 2d))
        self.t_off, ent_t, off = {}, [], 0
        for l in range(spec.layers):
            for key, (n, k) in (("qkv", (d, 3 * d)), ("attention.output.dense", (d, d)), ("intermediate.dense", (d, di)), ("output.dense", (di, d))) + (
                    (("kvg", (d, 2 * d)),) if rec else ()):
                self.t_off[(l, key)] = off
                ent_t.append((off, n, k))
                off += n * k
        self.scratch = torch.zeros(off, dtype=torch.float32, device=device)
        self.table_t = ops.KBlockTable(ent_t, device)
        self.fwd = self.bwd = None
        self.version = None

    def refresh(self, flat: torch.Tensor, version=None):
        if version is not None and version == self.version and self.fwd is not None:
            return self
        sp, p = self.spec, self.prefix
        d, di = sp.hidden, sp.intermediate
        w = self.layout.views(flat)
        self.fwd = ops.split_weights_kblock(flat, self.table, self.fwd)
        for l in range(sp.layers):
            lp = f"{p}encoder.layer.{l}."
            o = self.t_off[(l, "qkv")]
            wt = self.scratch[o:o + d * 3 * d].view(d, 3 * d)
            for k, n in enumerate(("query", "key", "value")):
                ops.transpose_pad(w[f"{lp}attention.self.{n}.weight"], out=wt[:, k * d:(k + 1) * d])
            for key, (r, c) in (("attention.output.dense", (d, d)), ("intermediate.dense", (d, di)), ("output.dense", (di, d))):
                o = self.t_off[(l, key)]
                ops.transpose_pad(w[lp + key + ".weight"], out=self.scratch[o:o + r * c].view(r, c))
            if sp.kind == "recformer":
                o = self.t_off[(l, "kvg")]
                wt = self.scratch[o:o + d * 2 * d].view(d, 2 * d)
                for k, n in enumerate(("key_global", "value_global")):
                    ops.transpose_pad(w[f"{lp}attention.self.{n}.weight"], out=wt[:, k * d:(k + 1) * d])
        self.bwd = ops.split_weights_kblock(self.scratch, self.table_t, self.bwd)
        self.version = version
        return self


class EncoderTrainGraph:
    """``mode``: "f32" -- every product through the exact-fp32 NT GEMM (merge_train's tiny batches; bit-comparable with the oracle);
    "bf16x3" -- the split-precision MFMA GEMM of the inference path (three bf16 products per fp32 product, ~1e-6 relative) for the
    token-sized batches of fine-tuning, with split-K weight gradients; needs ``split_weights`` (a refreshed SplitWeights)."""

    def __init__(self, spec: EncoderSpec, layout: ArenaLayout, prefix: str = "model.", mode: str = "f32", split_weights: Optional[SplitWeights] = None,
                 dropout: Optional[Dropout] = None):
        if spec.hidden // spec.heads != 64:
            raise ValueError("attention kernels are built for head_dim == 64")
        if mode not in ("f32", "bf16x3"):
            raise ValueError("training graph mode must be 'f32' or 'bf16x3'")
        if mode == "bf16x3" and (split_weights is None or spec.hidden % 128):
            raise ValueError("bf16x3 training needs SplitWeights and hidden % 128 == 0")
        self.spec, self.layout, self.prefix = spec, layout, prefix
        self.mode, self.sw = mode, split_weights
        self.rec = spec.kind == "recformer"
        self.window = spec.one_sided_window if self.rec else -1
        self.drop = dropout if (dropout is not None and dropout.active) else None  # None: the deterministic graph, bit for bit
        self._saved = None
        # persistent (buffer, named views) pairs of the caller for the flat parameter vector and for the gradient arena: 200-odd view objects
        # per vector and step are host time the 600-token step does not have (the step is launch-bound once the streams overlap)
        self.param_cache = None   # (tensor, views): used when forward()'s ``flat`` is that tensor's memory
        self.grad_cache = None    # (tensor, views): the gradient arena, zero-filled here at the start of every backward
        self.overlap = None  # a MergeOverlap: ``flat`` arrives range by range from a second stream; per-layer d alpha contractions in backward
        # arithmetic of the token-sized products in "f32" mode: bf16x6 (fp32-grade: ~2^-24 per product over fp32's whole range, gradients of
        # 1e-6 included; 2.7 x fewer matrix-pipe cycles than the fp32 MFMA) unless MR_TRAIN_TILE_PRODUCTS=0 asks for the exact FMA chain
        self.tile_products = _TILE_PRODUCTS

    def _views(self, flat: torch.Tensor):
        c = self.param_cache
        if c is not None and c[0].data_ptr() == flat.data_ptr() and c[0].numel() == flat.numel():
            return c[1]
        return self.layout.views(flat)

    def _grad_arena(self, like: torch.Tensor):
        c = self.grad_cache
        if c is not None and c[0].numel() == like.numel() and c[0].device == like.device:
            c[0].zero_()
            return c
        g_flat = torch.zeros_like(like)
        return g_flat, self.layout.views(g_flat)

    # ---------------------------------------------------------------------------------------------- products
    def _gt(self, *args, **kwargs):
        """ops.gemm_tile in this graph's token-sized arithmetic (``tile_products``: 0 = exact fp32 FMA chain, 6 = bf16x6 split precision)"""
        return ops.gemm_tile(*args, products=self.tile_products, **kwargs)

    def _linear(self, x, w, name: str, residual=None, out=None):
        """x W^T + b for the Linear ``name`` (arena key without ".weight")."""
        if self.mode == "f32" and _TILE:
            return self._gt(x, [w[name + ".weight"]], biases=[w[name + ".bias"]], residual=residual, out=out)
        if self.mode == "f32":
            return ops.gemm_nt_train(x, w[name + ".weight"], w[name + ".bias"], residual=residual, out=out)
        n, k = self.layout.shapes[name + ".weight"]
        return ops.gemm_nt_split(x, self.sw.fwd, [self.layout.offsets[name + ".weight"]], n, k, biases=[w[name + ".bias"]], residual=residual,
                                 out=out, products=3)

    def _tpad(self, x, out=None):
        """token-major transpose: (T, C) -> (C, T_pad), pad columns zero (T_pad % 32 == 0 in bf16x3 mode: whole prefetch pairs)"""
        return ops.transpose_pad(x, out=out, pad=32 if self.mode == "bf16x3" else 16)

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, flat: torch.Tensor, pb: PackedBatch) -> torch.Tensor:
        """-> (B, d) CLS rows of the last layer (not normalised); keeps what backward needs."""
        sp, p = self.spec, self.prefix
        w = self._views(flat)
        e = p + "embeddings."
        ov = self.overlap
        if ov is not None:
            ov.wait("others", first_only=True)  # the embedding range
        # pre-LayerNorm embedding sum (the fused inference kernel does not expose it): three row gathers
        emb = ops.gather_rows(w[e + "word_embeddings.weight"], pb.tok_word) + ops.gather_rows(w[e + "position_embeddings.weight"], pb.tok_pos)
        if self.rec:  # recformer/models.py:104-136: + token_type[tt] + item_position[ip]
            emb = emb + ops.gather_rows(w[e + "token_type_embeddings.weight"], pb.tok_tt) + ops.gather_rows(w[e + "item_position_embeddings.weight"], pb.tok_ip)
        else:
            emb = emb + w[e + "token_type_embeddings.weight"][0]
        x = ops.layernorm(emb, w[e + "LayerNorm.weight"], w[e + "LayerNorm.bias"], sp.ln_eps)
        dr = self.drop
        ph, pa = (dr.p_hidden, dr.p_attn) if dr else (0.0, 0.0)
        if ph > 0.0:
            ops.dropout_rows(x, ph, dr.key(0, ops.DROP_SITE_EMBED), out=x)
        saved = dict(pb=pb, flat=flat, emb=emb, layers=[])
        for l in range(sp.layers):
            lp = f"{p}encoder.layer.{l}."
            if ov is not None:
                ov.wait(str(l))
            names = [f"{lp}attention.self.{n}" for n in ("query", "key", "value")]
            qkv = torch.empty(pb.T, 3 * sp.hidden, dtype=torch.float32, device=x.device)
            if self.mode == "f32" and _TILE:  # one launch over the three weights (csrc/gemm_train.hip: no split-K, no reduce)
                self._gt(x, [w[n + ".weight"] for n in names], biases=[w[n + ".bias"] for n in names], out=qkv)
            elif self.mode == "f32":
                for s, n in enumerate(names):
                    ops.gemm_nt_train(x, w[n + ".weight"], w[n + ".bias"], out=qkv[:, s * sp.hidden:(s + 1) * sp.hidden])
            else:  # one launch over the three weight segments
                ops.gemm_nt_split(x, self.sw.fwd, [self.layout.offsets[n + ".weight"] for n in names], sp.hidden, sp.hidden,
                                  biases=[w[n + ".bias"] for n in names], out=qkv, products=3)
            # (bf16x3 mode: the split-precision attention of the inference path; its backward recomputes the probabilities in fp32)
            ctx = ops.attention(qkv, pb.cu_seqlens, pb.B, sp.heads, pb.max_len, window=self.window, seq_order=pb.seq_order,
                                products=3 if self.mode == "bf16x3" else 0, work=pb.attn_work,
                                drop_p=pa, drop_key=dr.key(l, ops.DROP_SITE_ATTN_PROBS) if pa > 0.0 else 0)
            qg = kvg = None
            if self.rec:  # Longformer global row: CLS attends to every token through the *_global projections and overwrites ctx[cls]
                x_cls = ops.gather_rows(x, pb.cls_rows)
                qg = self._linear(x_cls, w, f"{lp}attention.self.query_global")
                kvg = torch.empty(pb.T, 2 * sp.hidden, dtype=torch.float32, device=x.device)
                if self.mode == "f32" and _TILE:
                    gn = [f"{lp}attention.self.{n}" for n in ("key_global", "value_global")]
                    self._gt(x, [w[n + ".weight"] for n in gn], biases=[w[n + ".bias"] for n in gn], out=kvg)
                else:
                    for s, n in enumerate(("key_global", "value_global")):
                        self._linear(x, w, f"{lp}attention.self.{n}", out=kvg[:, s * sp.hidden:(s + 1) * sp.hidden])
                ops.attention_global_row(qg, kvg, pb.cu_seqlens, pb.B, sp.heads, pb.max_len, ctx,
                                         drop_p=pa, drop_key=dr.key(l, ops.DROP_SITE_GLOBAL_ROW) if pa > 0.0 else 0)
            if self.mode == "f32" and _TILE:  # the product's epilogue carries bias, dropout mask, residual and GELU: one launch per Linear
                n_o, n_1, n_2 = lp + "attention.output.dense", lp + "intermediate.dense", lp + "output.dense"
                a = self._gt(ctx, [w[n_o + ".weight"]], biases=[w[n_o + ".bias"]], residual=x, drop_p=ph,
                                  drop_key=dr.key(l, ops.DROP_SITE_ATTN_OUT) if ph > 0.0 else 0)
                h = ops.layernorm(a, w[lp + "attention.output.LayerNorm.weight"], w[lp + "attention.output.LayerNorm.bias"], sp.ln_eps)
                i = torch.empty(pb.T, sp.intermediate, dtype=torch.float32, device=x.device)
                u = self._gt(h, [w[n_1 + ".weight"]], biases=[w[n_1 + ".bias"]], epi=ops.EPI_GELU_FWD, out2=i)
                o = self._gt(i, [w[n_2 + ".weight"]], biases=[w[n_2 + ".bias"]], residual=h, drop_p=ph,
                                  drop_key=dr.key(l, ops.DROP_SITE_FFN_OUT) if ph > 0.0 else 0)
            else:
                if ph > 0.0:  # a = dropout(ctx Wo^T + bo) + x ; o = dropout(i W2^T + b2) + h: the residual joins after the mask
                    a = self._linear(ctx, w, lp + "attention.output.dense")
                    ops.dropout_rows(a, ph, dr.key(l, ops.DROP_SITE_ATTN_OUT), residual=x, out=a)
                else:
                    a = self._linear(ctx, w, lp + "attention.output.dense", residual=x)
                h = ops.layernorm(a, w[lp + "attention.output.LayerNorm.weight"], w[lp + "attention.output.LayerNorm.bias"], sp.ln_eps)
                u = self._linear(h, w, lp + "intermediate.dense")
                i = ops.gelu_fwd(u)
                if ph > 0.0:
                    o = self._linear(i, w, lp + "output.dense")
                    ops.dropout_rows(o, ph, dr.key(l, ops.DROP_SITE_FFN_OUT), residual=h, out=o)
                else:
                    o = self._linear(i, w, lp + "output.dense", residual=h)
            x_next = ops.layernorm(o, w[lp + "output.LayerNorm.weight"], w[lp + "output.LayerNorm.bias"], sp.ln_eps)
            saved["layers"].append(dict(x=x, qkv=qkv, ctx=ctx, a=a, h=h, u=u, i=i, o=o, qg=qg, kvg=kvg, l=l))
            x = x_next
        self._saved = saved
        if ov is not None:
            ov.wait()  # whatever is left (the pooler range): the vector is whole before anything outside this graph reads it
        return ops.gather_rows(x, pb.cls_rows)

    # ---------------------------------------------------------------------------------------------- backward
    def _dgrad(self, dy: torch.Tensor, w, l: int, key: str, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dX = dY @ W (+ residual) for layer l's Linear ``key`` ("qkv" / "kvg" = the stacked projections): the NT kernel on W^T."""
        sp, d = self.spec, self.spec.hidden
        if self.mode == "bf16x3":
            n, k = {"qkv": (d, 3 * d), "kvg": (d, 2 * d), "attention.output.dense": (d, d), "intermediate.dense": (d, sp.intermediate),
                    "output.dense": (sp.intermediate, d)}[key]
            return ops.gemm_nt_split(dy, self.sw.bwd, [self.sw.t_off[(l, key)]], n, k, residual=residual, products=3)
        lp = f"{self.prefix}encoder.layer.{l}."
        if key in ("qkv", "kvg"):
            names = ("query", "key", "value") if key == "qkv" else ("key_global", "value_global")
            wt = torch.empty(d, len(names) * d, dtype=torch.float32, device=dy.device)  # [Wq; Wk; Wv]^T
            for k, n in enumerate(names):
                ops.transpose_pad(w[f"{lp}attention.self.{n}.weight"], out=wt[:, k * d:(k + 1) * d])
            return ops.gemm_nt_train(dy, wt, residual=residual)
        return ops.gemm_nt_train(dy, ops.transpose_pad(w[lp + key + ".weight"]), residual=residual)

    def _xt(self, x: torch.Tensor):
        """the token-major operand of a weight gradient, prepared once per activation: f32 mode x^T (C, T_pad); bf16x3 mode the bf16
        pieces of x^T straight from x (no fp32 transpose)"""
        if self.mode == "bf16x3":
            pieces, t_pad = ops.split_tokens_kblock(x, pad=32)
            return (x.shape[1], t_pad), pieces
        return self._tpad(x), None

    def _wgrad(self, dy_t: torch.Tensor, xt, out: torch.Tensor):
        """dW = dY^T @ X written into ``out`` (a view of the gradient arena): both operands token-major transposed."""
        x_t, pieces = xt
        if pieces is None:
            ops.gemm_nt_train(dy_t, x_t, out=out)
        else:
            ops.gemm_nt_split_k(dy_t, pieces, 0, x_t[0], x_t[1], out=out)

    def _backward_f32(self, d_cls: torch.Tensor) -> torch.Tensor:
        """The exact-fp32 backward on the token-sized tile kernel (csrc/gemm_train.hip): every weight gradient reads dY and X as they lie
        (k = tokens down the rows), every input gradient reads W as it lies (k = output features down the rows) -- no transposed copy of
        anything -- and the epilogues carry what used to be separate launches: the bias gradient with its weight gradient, GELU' with
        the FFN input gradient, the residual adds.  Per layer: 8 products (+ 4 for Recformer's global projections) instead of 8 products
        + 8 split-K reductions + 14 transposes + 4 row sums + a GELU-backward launch."""
        sv = self._saved
        sp, p, pb = self.spec, self.prefix, sv["pb"]
        w = self._views(sv["flat"])
        g_flat, g = self._grad_arena(sv["flat"])
        d = sp.hidden
        dx = torch.zeros(pb.T, d, dtype=torch.float32, device=d_cls.device)
        ops.scatter_add_rows(d_cls.contiguous(), pb.cls_rows, dx)
        dr = self.drop
        ph, pa = (dr.p_hidden, dr.p_attn) if dr else (0.0, 0.0)
        T = True
        # The weight gradients feed nothing in this chain (they meet the rest of the step again in the alpha-gradient contraction), and a
        # 600-token product is one 12-wave workgroup per CU whose fill / drain / per-tile barriers leave the CU idle: they run on a second
        # stream beside the input-gradient chain.  Same launches, same operands, disjoint outputs: bit-identical.  Every tensor a
        # side-stream launch reads stays referenced until the main stream has waited for the side stream (allocator reuse).
        main = torch.cuda.current_stream(d_cls.device)
        ov = self.overlap
        side = (ov.stream if ov is not None else self._wgrad_stream(d_cls.device)) if (_WGRAD_STREAM or ov is not None) else None
        keep = []

        def wgrad(dy, xs, **kw):
            if side is None:
                return self._gt(dy, xs, trans_a=T, trans_b=T, **kw)
            side.wait_stream(main)  # everything issued so far: dy, the zeroed gradient arena
            with torch.cuda.stream(side):
                self._gt(dy, xs, trans_a=T, trans_b=T, **kw)
            keep.append((dy, xs))

        for l in reversed(range(sp.layers)):
            lp = f"{p}encoder.layer.{l}."
            s = sv["layers"][l]
            n_o, n_1, n_2 = lp + "attention.output.dense", lp + "intermediate.dense", lp + "output.dense"
            do = ops.layernorm_bwd(s["o"], dx, w[lp + "output.LayerNorm.weight"], sp.ln_eps, g[lp + "output.LayerNorm.weight"],
                                   g[lp + "output.LayerNorm.bias"])
            dod = ops.dropout_rows(do, ph, dr.key(l, ops.DROP_SITE_FFN_OUT)) if ph > 0.0 else do   # the dense sees the masked gradient
            wgrad(dod, [s["i"]], out=[g[n_2 + ".weight"]], colsum=[g[n_2 + ".bias"]])       # dW2 = dY^T i, db2
            du = self._gt(dod, [w[n_2 + ".weight"]], trans_b=T, epi=ops.EPI_GELU_BWD, E=s["u"])                      # (dY W2) * gelu'(u)
            wgrad(du, [s["h"]], out=[g[n_1 + ".weight"]], colsum=[g[n_1 + ".bias"]])         # dW1, db1
            dh = self._gt(du, [w[n_1 + ".weight"]], trans_b=T, residual=do)                                          # + the residual path of o
            da = ops.layernorm_bwd(s["a"], dh, w[lp + "attention.output.LayerNorm.weight"], sp.ln_eps,
                                   g[lp + "attention.output.LayerNorm.weight"], g[lp + "attention.output.LayerNorm.bias"])
            dad = ops.dropout_rows(da, ph, dr.key(l, ops.DROP_SITE_ATTN_OUT)) if ph > 0.0 else da
            wgrad(dad, [s["ctx"]], out=[g[n_o + ".weight"]], colsum=[g[n_o + ".bias"]])     # dWo, dbo
            dctx = self._gt(dad, [w[n_o + ".weight"]], trans_b=T)
            dqkv = ops.attention_bwd(s["qkv"], s["ctx"], dctx, pb.cu_seqlens, pb.B, sp.heads, window=self.window, max_len=pb.max_len, seq_order=pb.seq_order,
                                     drop_p=pa, drop_key=dr.key(l, ops.DROP_SITE_ATTN_PROBS) if pa > 0.0 else 0, work=pb.attn_work)
            if self.rec:
                dqg, dkvg = ops.attention_global_row_bwd(s["qg"], s["kvg"], ops.gather_rows(s["ctx"], pb.cls_rows), ops.gather_rows(dctx, pb.cls_rows),
                                                         pb.cu_seqlens, pb.B, sp.heads,
                                                         drop_p=pa, drop_key=dr.key(l, ops.DROP_SITE_GLOBAL_ROW) if pa > 0.0 else 0)
                gn = [f"{lp}attention.self.{n}" for n in ("key_global", "value_global")]
                wgrad(dkvg, [s["x"]], out=[g[n + ".weight"] for n in gn], colsum=[g[n + ".bias"] for n in gn])
                da = self._gt(dkvg, [w[n + ".weight"] for n in gn], trans_b=T, residual=da)
                # query_global reads the CLS rows only
                name = f"{lp}attention.self.query_global"
                x_cls = ops.gather_rows(s["x"], pb.cls_rows)
                wgrad(dqg, [x_cls], out=[g[name + ".weight"]], colsum=[g[name + ".bias"]])
                ops.scatter_add_rows(self._gt(dqg, [w[name + ".weight"]], trans_b=T), pb.cls_rows, da)
            qn = [f"{lp}attention.self.{n}" for n in ("query", "key", "value")]
            wgrad(dqkv, [s["x"]], out=[g[n + ".weight"] for n in qn], colsum=[g[n + ".bias"] for n in qn])
            dx = self._gt(dqkv, [w[n + ".weight"] for n in qn], trans_b=T, residual=da)                               # + the residual path of a
            if ov is not None:
                ov.contract(str(l), g_flat)  # this layer's gradients are complete: its d alpha range, under the next layer's backward
        e = p + "embeddings."
        if ph > 0.0:  # x0 = dropout(LN(emb))
            dx = ops.dropout_rows(dx, ph, dr.key(0, ops.DROP_SITE_EMBED), out=dx)
        de = ops.layernorm_bwd(sv["emb"], dx, w[e + "LayerNorm.weight"], sp.ln_eps, g[e + "LayerNorm.weight"], g[e + "LayerNorm.bias"])
        ops.scatter_add_rows(de, pb.tok_word, g[e + "word_embeddings.weight"])
        ops.scatter_add_rows(de, pb.tok_pos, g[e + "position_embeddings.weight"])
        if self.rec:
            ops.scatter_add_rows(de, pb.tok_tt, g[e + "token_type_embeddings.weight"])
            ops.scatter_add_rows(de, pb.tok_ip, g[e + "item_position_embeddings.weight"])
        else:
            ops.colsum(de, g[e + "token_type_embeddings.weight"][0])
        if ov is not None:
            ov.contract("others", g_flat, d_emb=de)  # embeddings (complete only now) and pooler
        if side is not None:
            main.wait_stream(side)  # the gradient arena is complete for whatever the main stream runs next
        del keep
        self._saved = None
        return g_flat

    def _wgrad_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None:
            st = self._side = torch.cuda.Stream(device=device)
        return st

    def backward(self, d_cls: torch.Tensor) -> torch.Tensor:
        """d loss / d CLS rows (B, d) -> d loss / d parameters, flat, arena layout (pads zero)."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("backward() without a preceding forward()")
        if self.mode == "f32" and _TILE:
            return self._backward_f32(d_cls)
        sp, p, pb = self.spec, self.prefix, sv["pb"]
        w = self._views(sv["flat"])
        g_flat, g = self._grad_arena(sv["flat"])
        d = sp.hidden
        dx = torch.zeros(pb.T, d, dtype=torch.float32, device=d_cls.device)
        ops.scatter_add_rows(d_cls.contiguous(), pb.cls_rows, dx)
        dr = self.drop
        ph, pa = (dr.p_hidden, dr.p_attn) if dr else (0.0, 0.0)
        for l in reversed(range(sp.layers)):
            lp = f"{p}encoder.layer.{l}."
            s = sv["layers"][l]
            # x_next = LN2(o)
            do = ops.layernorm_bwd(s["o"], dx, w[lp + "output.LayerNorm.weight"], sp.ln_eps, g[lp + "output.LayerNorm.weight"],
                                   g[lp + "output.LayerNorm.bias"])
            # o = dropout(i W2^T + b2) + h: the dense sees the masked gradient, the residual path (below) the whole one
            dod = ops.dropout_rows(do, ph, dr.key(l, ops.DROP_SITE_FFN_OUT)) if ph > 0.0 else do
            do_t = self._tpad(dod)
            ops.rowsum(do_t, g[lp + "output.dense.bias"])  # bias gradient = row sums of dY^T
            self._wgrad(do_t, self._xt(s["i"]), g[lp + "output.dense.weight"])
            di = self._dgrad(dod, w, l, "output.dense")
            # i = gelu(u), u = h W1^T + b1
            du = ops.gelu_bwd(s["u"], di)
            du_t = self._tpad(du)
            ops.rowsum(du_t, g[lp + "intermediate.dense.bias"])
            self._wgrad(du_t, self._xt(s["h"]), g[lp + "intermediate.dense.weight"])
            dh = self._dgrad(du, w, l, "intermediate.dense", residual=do)  # + the residual path of o
            # h = LN1(a)
            da = ops.layernorm_bwd(s["a"], dh, w[lp + "attention.output.LayerNorm.weight"], sp.ln_eps,
                                   g[lp + "attention.output.LayerNorm.weight"], g[lp + "attention.output.LayerNorm.bias"])
            # a = dropout(ctx Wo^T + bo) + x
            dad = ops.dropout_rows(da, ph, dr.key(l, ops.DROP_SITE_ATTN_OUT)) if ph > 0.0 else da
            da_t = self._tpad(dad)
            ops.rowsum(da_t, g[lp + "attention.output.dense.bias"])
            self._wgrad(da_t, self._xt(s["ctx"]), g[lp + "attention.output.dense.weight"])
            dctx = self._dgrad(dad, w, l, "attention.output.dense")
            dqkv = ops.attention_bwd(s["qkv"], s["ctx"], dctx, pb.cu_seqlens, pb.B, sp.heads, window=self.window, max_len=pb.max_len, seq_order=pb.seq_order,
                                     drop_p=pa, drop_key=dr.key(l, ops.DROP_SITE_ATTN_PROBS) if pa > 0.0 else 0, work=pb.attn_work)
            xt = self._xt(s["x"])
            if self.rec:
                dqg, dkvg = ops.attention_global_row_bwd(s["qg"], s["kvg"], ops.gather_rows(s["ctx"], pb.cls_rows), ops.gather_rows(dctx, pb.cls_rows),
                                                         pb.cu_seqlens, pb.B, sp.heads,
                                                         drop_p=pa, drop_key=dr.key(l, ops.DROP_SITE_GLOBAL_ROW) if pa > 0.0 else 0)
                # key_global / value_global read every token
                dkvg_t = self._tpad(dkvg)
                bs2 = ops.rowsum(dkvg_t)
                for k, n in enumerate(("key_global", "value_global")):
                    name = f"{lp}attention.self.{n}"
                    g[name + ".bias"].copy_(bs2[k * d:(k + 1) * d])
                    self._wgrad(dkvg_t[k * d:(k + 1) * d], xt, g[name + ".weight"])
                da = self._dgrad(dkvg, w, l, "kvg", residual=da)  # folded into the residual that the qkv d-grad below carries on
                # query_global reads the CLS rows only (a handful of rows: the exact-fp32 kernel in either mode)
                name = f"{lp}attention.self.query_global"
                ops.colsum(dqg, g[name + ".bias"])
                ops.gemm_nt_train(ops.transpose_pad(dqg), ops.transpose_pad(ops.gather_rows(s["x"], pb.cls_rows)), out=g[name + ".weight"])
                ops.scatter_add_rows(ops.gemm_nt_train(dqg, ops.transpose_pad(w[name + ".weight"])), pb.cls_rows, da)
            # qkv = x [Wq; Wk; Wv]^T + b
            dqkv_t = self._tpad(dqkv)  # (3 d, T_pad)
            bsum = ops.rowsum(dqkv_t)
            for k, n in enumerate(("query", "key", "value")):
                name = f"{lp}attention.self.{n}"
                g[name + ".bias"].copy_(bsum[k * d:(k + 1) * d])
                self._wgrad(dqkv_t[k * d:(k + 1) * d], xt, g[name + ".weight"])
            dx = self._dgrad(dqkv, w, l, "qkv", residual=da)  # + the residual path of a
        # x0 = LN(emb), emb = word[ids] + pos[pos_ids] + type[0]
        e = p + "embeddings."
        if ph > 0.0:  # x0 = dropout(LN(emb))
            dx = ops.dropout_rows(dx, ph, dr.key(0, ops.DROP_SITE_EMBED), out=dx)
        de = ops.layernorm_bwd(sv["emb"], dx, w[e + "LayerNorm.weight"], sp.ln_eps, g[e + "LayerNorm.weight"], g[e + "LayerNorm.bias"])
        ops.scatter_add_rows(de, pb.tok_word, g[e + "word_embeddings.weight"])
        ops.scatter_add_rows(de, pb.tok_pos, g[e + "position_embeddings.weight"])
        if self.rec:
            ops.scatter_add_rows(de, pb.tok_tt, g[e + "token_type_embeddings.weight"])
            ops.scatter_add_rows(de, pb.tok_ip, g[e + "item_position_embeddings.weight"])
        else:
            ops.colsum(de, g[e + "token_type_embeddings.weight"][0])
        self._saved = None
        return g_flat


RobertaTrainGraph = EncoderTrainGraph  # first name of the class (BLaIR only at the time)


class _EncodeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, graph: EncoderTrainGraph, pb: PackedBatch):
        ctx.graph = graph
        return graph.forward(flat, pb)

    @staticmethod
    def backward(ctx, d_cls):
        return ctx.graph.backward(d_cls.contiguous()), None, None


def encode_with_grad(graph: EncoderTrainGraph, flat: torch.Tensor, pb: PackedBatch) -> torch.Tensor:
    """(B, d) CLS rows with an autograd edge to the flat parameter vector ``flat`` (arena layout)."""
    return _EncodeFn.apply(flat, graph, pb)
