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

from typing import Optional

import torch

from . import ops
from .engine import ArenaLayout, EncoderSpec, PackedBatch

__all__ = ["EncoderTrainGraph", "RobertaTrainGraph", "encode_with_grad"]


class EncoderTrainGraph:
    def __init__(self, spec: EncoderSpec, layout: ArenaLayout, prefix: str = "model."):
        if spec.hidden // spec.heads != 64:
            raise ValueError("attention kernels are built for head_dim == 64")
        self.spec, self.layout, self.prefix = spec, layout, prefix
        self.rec = spec.kind == "recformer"
        self.window = spec.one_sided_window if self.rec else -1
        self._saved = None

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, flat: torch.Tensor, pb: PackedBatch) -> torch.Tensor:
        """-> (B, d) CLS rows of the last layer (not normalised); keeps what backward needs."""
        sp, p = self.spec, self.prefix
        w = self.layout.views(flat)
        e = p + "embeddings."
        # pre-LayerNorm embedding sum (the fused inference kernel does not expose it): three row gathers
        emb = ops.gather_rows(w[e + "word_embeddings.weight"], pb.tok_word) + ops.gather_rows(w[e + "position_embeddings.weight"], pb.tok_pos)
        if self.rec:  # recformer/models.py:104-136: + token_type[tt] + item_position[ip]
            emb = emb + ops.gather_rows(w[e + "token_type_embeddings.weight"], pb.tok_tt) + ops.gather_rows(w[e + "item_position_embeddings.weight"], pb.tok_ip)
        else:
            emb = emb + w[e + "token_type_embeddings.weight"][0]
        x = ops.layernorm(emb, w[e + "LayerNorm.weight"], w[e + "LayerNorm.bias"], sp.ln_eps)
        saved = dict(pb=pb, flat=flat, emb=emb, layers=[])
        for l in range(sp.layers):
            lp = f"{p}encoder.layer.{l}."
            names = [f"{lp}attention.self.{n}" for n in ("query", "key", "value")]
            qkv = torch.empty(pb.T, 3 * sp.hidden, dtype=torch.float32, device=x.device)
            for s, n in enumerate(names):
                ops.gemm_nt_train(x, w[n + ".weight"], w[n + ".bias"], out=qkv[:, s * sp.hidden:(s + 1) * sp.hidden])
            ctx = ops.attention(qkv, pb.cu_seqlens, pb.B, sp.heads, pb.max_len, window=self.window, seq_order=pb.seq_order, products=0)
            qg = kvg = None
            if self.rec:  # Longformer global row: CLS attends to every token through the *_global projections and overwrites ctx[cls]
                x_cls = ops.gather_rows(x, pb.cls_rows)
                qg = ops.gemm_nt_train(x_cls, w[f"{lp}attention.self.query_global.weight"], w[f"{lp}attention.self.query_global.bias"])
                kvg = torch.empty(pb.T, 2 * sp.hidden, dtype=torch.float32, device=x.device)
                for s, n in enumerate(("key_global", "value_global")):
                    ops.gemm_nt_train(x, w[f"{lp}attention.self.{n}.weight"], w[f"{lp}attention.self.{n}.bias"], out=kvg[:, s * sp.hidden:(s + 1) * sp.hidden])
                ops.attention_global_row(qg, kvg, pb.cu_seqlens, pb.B, sp.heads, pb.max_len, ctx)
            a = ops.gemm_nt_train(ctx, w[lp + "attention.output.dense.weight"], w[lp + "attention.output.dense.bias"], residual=x)
            h = ops.layernorm(a, w[lp + "attention.output.LayerNorm.weight"], w[lp + "attention.output.LayerNorm.bias"], sp.ln_eps)
            u = ops.gemm_nt_train(h, w[lp + "intermediate.dense.weight"], w[lp + "intermediate.dense.bias"])
            i = ops.gelu_fwd(u)
            o = ops.gemm_nt_train(i, w[lp + "output.dense.weight"], w[lp + "output.dense.bias"], residual=h)
            x_next = ops.layernorm(o, w[lp + "output.LayerNorm.weight"], w[lp + "output.LayerNorm.bias"], sp.ln_eps)
            saved["layers"].append(dict(x=x, qkv=qkv, ctx=ctx, a=a, h=h, u=u, i=i, o=o, qg=qg, kvg=kvg))
            x = x_next
        self._saved = saved
        return ops.gather_rows(x, pb.cls_rows)

    # ---------------------------------------------------------------------------------------------- backward
    @staticmethod
    def _dgrad(dy: torch.Tensor, W: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dX = dY @ W (+ residual): the NT kernel on W^T."""
        return ops.gemm_nt_train(dy, ops.transpose_pad(W), residual=residual)

    @staticmethod
    def _wgrad(dy_t: torch.Tensor, x_t: torch.Tensor, out: torch.Tensor):
        """dW = dY^T @ X written into ``out`` (a view of the gradient arena): both operands token-major transposed."""
        ops.gemm_nt_train(dy_t, x_t, out=out)

    def backward(self, d_cls: torch.Tensor) -> torch.Tensor:
        """d loss / d CLS rows (B, d) -> d loss / d parameters, flat, arena layout (pads zero)."""
        sv = self._saved
        if sv is None:
            raise RuntimeError("backward() without a preceding forward()")
        sp, p, pb = self.spec, self.prefix, sv["pb"]
        w = self.layout.views(sv["flat"])
        g_flat = torch.zeros_like(sv["flat"])
        g = self.layout.views(g_flat)
        d = sp.hidden
        dx = torch.zeros(pb.T, d, dtype=torch.float32, device=d_cls.device)
        ops.scatter_add_rows(d_cls.contiguous(), pb.cls_rows, dx)
        for l in reversed(range(sp.layers)):
            lp = f"{p}encoder.layer.{l}."
            s = sv["layers"][l]
            # x_next = LN2(o)
            do = ops.layernorm_bwd(s["o"], dx, w[lp + "output.LayerNorm.weight"], sp.ln_eps, g[lp + "output.LayerNorm.weight"],
                                   g[lp + "output.LayerNorm.bias"])
            # o = i W2^T + b2 + h
            do_t = ops.transpose_pad(do)
            ops.colsum(do, g[lp + "output.dense.bias"])
            self._wgrad(do_t, ops.transpose_pad(s["i"]), g[lp + "output.dense.weight"])
            di = self._dgrad(do, w[lp + "output.dense.weight"])
            # i = gelu(u), u = h W1^T + b1
            du = ops.gelu_bwd(s["u"], di)
            ops.colsum(du, g[lp + "intermediate.dense.bias"])
            h_t = ops.transpose_pad(s["h"])
            self._wgrad(ops.transpose_pad(du), h_t, g[lp + "intermediate.dense.weight"])
            dh = self._dgrad(du, w[lp + "intermediate.dense.weight"], residual=do)  # + the residual path of o
            # h = LN1(a)
            da = ops.layernorm_bwd(s["a"], dh, w[lp + "attention.output.LayerNorm.weight"], sp.ln_eps,
                                   g[lp + "attention.output.LayerNorm.weight"], g[lp + "attention.output.LayerNorm.bias"])
            # a = ctx Wo^T + bo + x
            ops.colsum(da, g[lp + "attention.output.dense.bias"])
            self._wgrad(ops.transpose_pad(da), ops.transpose_pad(s["ctx"]), g[lp + "attention.output.dense.weight"])
            dctx = self._dgrad(da, w[lp + "attention.output.dense.weight"])
            dqkv = ops.attention_bwd(s["qkv"], s["ctx"], dctx, pb.cu_seqlens, pb.B, sp.heads, window=self.window)
            x_t = ops.transpose_pad(s["x"])
            if self.rec:
                dqg, dkvg = ops.attention_global_row_bwd(s["qg"], s["kvg"], ops.gather_rows(s["ctx"], pb.cls_rows), ops.gather_rows(dctx, pb.cls_rows),
                                                         pb.cu_seqlens, pb.B, sp.heads)
                # key_global / value_global read every token
                dkvg_t = ops.transpose_pad(dkvg)
                bs2 = ops.colsum(dkvg)
                wt2 = torch.empty(d, 2 * d, dtype=torch.float32, device=dx.device)
                for k, n in enumerate(("key_global", "value_global")):
                    name = f"{lp}attention.self.{n}"
                    g[name + ".bias"].copy_(bs2[k * d:(k + 1) * d])
                    self._wgrad(dkvg_t[k * d:(k + 1) * d], x_t, g[name + ".weight"])
                    ops.transpose_pad(w[name + ".weight"], out=wt2[:, k * d:(k + 1) * d])
                da = ops.gemm_nt_train(dkvg, wt2, residual=da)  # folded into the residual that the qkv d-grad below carries on
                # query_global reads the CLS rows only
                name = f"{lp}attention.self.query_global"
                ops.colsum(dqg, g[name + ".bias"])
                self._wgrad(ops.transpose_pad(dqg), ops.transpose_pad(ops.gather_rows(s["x"], pb.cls_rows)), g[name + ".weight"])
                ops.scatter_add_rows(self._dgrad(dqg, w[name + ".weight"]), pb.cls_rows, da)
            # qkv = x [Wq; Wk; Wv]^T + b
            dqkv_t = ops.transpose_pad(dqkv)  # (3 d, T_pad)
            bsum = ops.colsum(dqkv)
            wt = torch.empty(d, 3 * d, dtype=torch.float32, device=dx.device)  # [Wq; Wk; Wv]^T
            for k, n in enumerate(("query", "key", "value")):
                name = f"{lp}attention.self.{n}"
                g[name + ".bias"].copy_(bsum[k * d:(k + 1) * d])
                self._wgrad(dqkv_t[k * d:(k + 1) * d], x_t, g[name + ".weight"])
                ops.transpose_pad(w[name + ".weight"], out=wt[:, k * d:(k + 1) * d])
            dx = ops.gemm_nt_train(dqkv, wt, residual=da)  # + the residual path of a
        # x0 = LN(emb), emb = word[ids] + pos[pos_ids] + type[0]
        e = p + "embeddings."
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
