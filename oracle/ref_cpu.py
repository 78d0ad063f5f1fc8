"""CPU oracle for the MergeRec merged-model inference path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it, and there only
as the checker / the timed CPU baseline -- never as the thing shipped.  The product path
(``mergerec_amd``) calls hand-written HIP kernels through ``libmergerec_hip.so`` and fails loudly
when that library is missing.

This file is a plain fp32 ``torch`` (CPU) restatement of the reference's arithmetic for the path
named by BASELINE.json's ``north_star``; every function cites the reference file:line it follows
(paths relative to the upstream repo root).  It does not import ``transformers``, ``lightning``,
``tyro`` or anything from the reference, so it travels to the GPU box.

Pinning ("parity pinned"): ``oracle/gen_golden.py`` (run in the build container only) imports the
runnable subset of the reference (``rec_retrieval.evaluator`` as-is; ``rec_retrieval.merger``
algorithms; the reference's own ``RecformerEmbeddings``/mask helpers) plus the third-party library
the reference delegates the encoder arithmetic to (``transformers`` RobertaModel / LongformerEncoder,
reference pins ~=4.51.3, container has 5.15.0) and writes small input/output vectors to
``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every function here against them.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

StateDict = Dict[str, torch.Tensor]

# --------------------------------------------------------------------------------------------
# a1..a3: checkpoint key handling, flatten, task vectors
# --------------------------------------------------------------------------------------------


def remove_duplicate_prefix(state_dict: StateDict) -> StateDict:
    """utils.py:17-29 -- strip the first ``"model."`` from every key that has it."""
    out = {}
    for k, v in state_dict.items():
        out[k.replace("model.", "", 1) if k.startswith("model.") else k] = v
    return out


def align_state_dicts(
    pretrain: StateDict, finetunes: Sequence[StateDict], ignore_keys: Sequence[str] = ()
) -> Tuple[StateDict, List[StateDict]]:
    """merger/weight_learning/module/_factory.py:55-66 -- keep ``pre ∩ ft[0]`` minus ignored, ordered
    by the PRETRAINED dict's insertion order; fine-tuned dicts re-ordered to match."""
    keep = (set(pretrain.keys()) & set(finetunes[0].keys())) - set(ignore_keys)
    pre = OrderedDict((k, v) for k, v in pretrain.items() if k in keep)
    fts = [OrderedDict((k, ft[k]) for k in pre.keys()) for ft in finetunes]
    return pre, fts


def flatten_model(model: StateDict) -> Tuple[torch.Tensor, "OrderedDict[str, torch.Size]"]:
    """merger/utils/model_operations.py:47-63 -- ``torch.cat([v.reshape(-1)])`` in dict order.
    ``torch.cat`` type-promotes, so an int64 buffer (Recformer ``position_ids``) becomes fp32."""
    shape_dict = OrderedDict((k, v.shape) for k, v in model.items())
    flat = torch.cat([v.reshape(-1) for v in model.values()])
    return flat, shape_dict


def get_task_vectors(base: torch.Tensor, models: Sequence[torch.Tensor]) -> torch.Tensor:
    """merger/algorithms/task_vector.py:8-10 -- ``stack([m - base])`` -> (N, P)."""
    return torch.stack([m - base for m in models])


def get_state_dict(params: torch.Tensor, shape_dict) -> StateDict:
    """merger/weight_learning/utils.py:29-40 -- slice the flat vector into named views."""
    out, start = OrderedDict(), 0
    for name, shape in shape_dict.items():
        n = math.prod(shape)
        out[name] = params[start : start + n].reshape(shape)
        start += n
    assert start == params.numel(), "Not all parameters are loaded."
    return out


# --------------------------------------------------------------------------------------------
# a4/a5: the merge
# --------------------------------------------------------------------------------------------


def effective_alpha(gw: torch.Tensor, gb: torch.Tensor, per: torch.Tensor, disable_softmax: bool) -> torch.Tensor:
    """task_wise.py:37-42 / layer_wise.py:67-73 -- ``alpha = gw * (softmax?)(per) + gb``."""
    if not disable_softmax:
        per = torch.softmax(per, dim=0)
    return gw * per + gb


def merge_task_wise(base: torch.Tensor, tv: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """task_wise.py:43-47 -- ``base + (alpha[:, None] * T).sum(0)`` (products rounded, then a
    sequential i=0..N-1 sum from 0, then ``base + sum``; no fused multiply-add)."""
    return base + (alpha.unsqueeze(1) * tv).sum(dim=0)


def merge_running(base: Optional[torch.Tensor], models: Sequence[torch.Tensor], weights: Sequence[float]) -> torch.Tensor:
    """ModelMerger.merge("task_vector") (merger/algorithms/task_vector.py:13-34: ``merged = base.clone(); merged += w_i * (m_i - base)``)
    and ("linear") (algorithms/linear.py:8-27: ``merged = zeros; merged += w_i * m_i``) -- running sums in model order."""
    assert len(models) == len(weights), "Number of models and weights should match."
    if base is None:
        merged = torch.zeros_like(models[0])
        for w, m in zip(weights, models):
            merged += w * m
        return merged
    merged = base.clone()
    for w, m in zip(weights, models):
        merged += w * (m - base)
    return merged


def group_parameters_by_layer(shape_dict) -> "OrderedDict[str, List[Tuple[str, int, int]]]":
    """layer_wise.py:13-33 -- group id = ``name.split('.')[3]`` when ``'encoder.layer.'`` is in the
    name, else ``'others'``; groups keep first-seen order."""
    groups: "OrderedDict[str, list]" = OrderedDict()
    off = 0
    for name, shape in shape_dict.items():
        n = math.prod(shape)
        key = name.split(".")[3] if "encoder.layer." in name else "others"
        groups.setdefault(key, []).append((name, off, off + n))
        off += n
    return groups


def merge_layer_wise(base: torch.Tensor, tv: torch.Tensor, groups, alpha_by_group: Dict[str, torch.Tensor]) -> torch.Tensor:
    """layer_wise.py:64-83 -- the task-wise expression applied per (name, start, end) chunk with
    the chunk's group coefficients, written into a zero-initialised vector."""
    merged = torch.zeros_like(base)
    for key, chunks in groups.items():
        a = alpha_by_group[key]
        for _, s, e in chunks:
            merged[s:e] = base[s:e] + (a.unsqueeze(1) * tv[:, s:e]).sum(dim=0)
    return merged


def merge_bwd_alpha(tv: torch.Tensor, grad: torch.Tensor) -> torch.Tensor:
    """Backward of task_wise.py:43-47 w.r.t. alpha: ``dalpha_i = <tau_i, g>`` (fp64 accumulate so
    the oracle is an accuracy anchor for the device's two-stage fp32 reduction)."""
    return (tv.double() @ grad.double()).float()


# --------------------------------------------------------------------------------------------
# 8(f).1: task-vector pre-processing run once at init (TIES, Localize-and-Stitch, PCB)
# --------------------------------------------------------------------------------------------


def topk_abs_mask(x: torch.Tensor, k: int) -> torch.Tensor:
    """Boolean mask of the k largest |x| (ties at the threshold resolved towards LOWER indices -- torch.topk leaves
    that order unspecified; algorithms/ties.py:21, localize_and_stitch.py:40)."""
    if k <= 0:
        return torch.zeros_like(x, dtype=torch.bool)
    order = torch.sort(x.abs(), descending=True, stable=True).indices[:k]
    m = torch.zeros_like(x, dtype=torch.bool)
    m[order] = True
    return m


def ties_vectors(base: torch.Tensor, models: Sequence[torch.Tensor], density: float) -> torch.Tensor:
    """algorithms/ties.py:8-72 -- per-model top-(density*P) by |tau|, sign election by summed mass, disjoint mean."""
    k = int(density * base.numel())
    sparse = []
    for m in models:
        u = m - base
        keep = topk_abs_mask(u, k)
        sparse.append(torch.where(keep, u, torch.zeros_like(u)))
    sp = torch.stack(sparse, 0)
    zero = torch.zeros_like(sp)
    pos_sum = torch.where(sp > 0, sp, zero).sum(0)
    neg_sum = torch.where(sp < 0, sp, zero).sum(0)
    conflict = (pos_sum != 0) & (neg_sum != 0)
    sign = torch.where(conflict, torch.where(pos_sum.abs() >= neg_sum.abs(), 1.0, -1.0), torch.sign(pos_sum + neg_sum))
    sign = torch.where(sign == 0, torch.ones_like(sign), sign)
    sel = torch.where(sign.unsqueeze(0) > 0, torch.where(sp > 0, sp, zero), torch.where(sp < 0, sp, zero))
    cnt = torch.count_nonzero(sel, dim=0).unsqueeze(0)
    return (sel / cnt).nan_to_num(0.0)


def localize_and_stitch_vectors(base: torch.Tensor, models: Sequence[torch.Tensor], density: float) -> torch.Tensor:
    """algorithms/localize_and_stitch.py:8-49 -- top-k masks, overlap-normalised: (mask_i / max(sum_j mask_j, 1)) * tau_i."""
    upd = torch.stack([m - base for m in models], 0)
    k = int(density * upd.shape[1])
    if k <= 0:
        return torch.zeros_like(upd)
    masks = torch.stack([topk_abs_mask(u, k) for u in upd], 0).to(upd.dtype)
    denom = masks.sum(0).clamp(min=1.0)
    return (masks / denom) * upd


def pcb_vectors(base: torch.Tensor, models: Sequence[torch.Tensor], density: float = 0.2) -> torch.Tensor:
    """algorithms/pcb.py:8-58."""
    tv = torch.stack([m - base for m in models], 0)
    n, d = tv.shape

    def clamp_rows(x, lo, hi):
        srt = torch.sort(x, dim=1).values
        return torch.maximum(torch.minimum(x, srt[:, int(d * (1 - hi) - 1)].unsqueeze(1)), srt[:, int(d * lo)].unsqueeze(1))

    def normalize(x):
        mn, mx = x.min(1, keepdim=True).values, x.max(1, keepdim=True).values
        return (x - mn) / (mx - mn)

    a = clamp_rows(tv.abs(), 0.01, 0.01)
    clamped = torch.sign(tv) * a
    self_act = torch.exp(n * normalize(a) ** 2)
    cross_act = torch.tanh(tv * tv.sum(0))
    scale = normalize(clamp_rows(self_act * cross_act, 1 - density, 0))
    out = clamped * scale
    out = out / torch.clamp(scale.sum(0, keepdim=True), min=1e-12)
    return out / n


# --------------------------------------------------------------------------------------------
# a8/a11/a13/a14: RoBERTa (BLaIR) encoder; the reference delegates to transformers RobertaModel
# via module/models/encoder/_base.py:32-39.  Restated from the library's published architecture.
# --------------------------------------------------------------------------------------------


# --------------------------------------------------------------------------------------------
# Training-graph dropout.  The reference trains under lightning.Trainer.fit, i.e. train() mode with torch's dropout at HF's sites
# (transformers RobertaEmbeddings / RobertaSelfAttention / RobertaSelfOutput / RobertaOutput and their Longformer twins;
# recformer/models.py:93,135) at HF's default rates 0.1 / 0.1 (merge_train.py:178-196).  torch's Philox stream cannot be restated outside
# torch, so the BUILD defines its mask as a pure function of (seed, step, layer, site, row, column) -- mergerec_amd/csrc/dropout.h --
# and this is its restatement: with the same plan the oracle's autograd step and the HIP step must agree to rounding.
# --------------------------------------------------------------------------------------------
DROP_SITE_EMBED, DROP_SITE_ATTN_PROBS, DROP_SITE_ATTN_OUT, DROP_SITE_FFN_OUT, DROP_SITE_GLOBAL_ROW = range(5)


def _lowbias32(x):
    """uint32 mixer (numpy uint64 arithmetic masked to 32 bits)."""
    import numpy as np

    m = np.uint64(0xFFFFFFFF)
    x = np.asarray(x, dtype=np.uint64) & m
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & m
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & m
    x ^= x >> np.uint64(16)
    return x


def dropout_site_key(seed: int, step: int, layer: int, site: int) -> int:
    return int(_lowbias32(int(_lowbias32(int(_lowbias32(seed)) + step)) + layer * 8 + site))


def dropout_keep(key: int, rows: torch.Tensor, cols: torch.Tensor, p: float) -> torch.Tensor:
    """keep[...] = lowbias32(row * 0x9E3779B1 + col * 0x85EBCA77 + key) >= floor(p * 2^32); rows / cols broadcast against each other."""
    import numpy as np

    m = np.uint64(0xFFFFFFFF)
    r = rows.to(torch.int64).numpy().astype(np.uint64)
    c = cols.to(torch.int64).numpy().astype(np.uint64)
    x = ((r * np.uint64(0x9E3779B1)) & m) + ((c * np.uint64(0x85EBCA77)) & m) + np.uint64(key)
    thresh = min(int(float(np.float32(p)) * 4294967296.0), 0xFFFFFFFF)
    return torch.from_numpy(_lowbias32(x & m) >= np.uint64(thresh))


def dropout_scale(p: float) -> torch.Tensor:
    """1 / (1 - p) in fp32, as the kernels compute it."""
    return torch.tensor(1.0, dtype=torch.float32) / (torch.tensor(1.0, dtype=torch.float32) - torch.tensor(p, dtype=torch.float32))


@dataclass
class DropoutPlan:
    """One training forward's dropout (== mergerec_amd.engine_train.Dropout).  ``tok`` (B, L) int64 = packed token index of every
    padded position (cumulative lengths + position), set by the encoders from the attention mask."""

    p_hidden: float = 0.1
    p_attn: float = 0.1
    seed: int = 0
    step: int = 0
    tok: Optional[torch.Tensor] = None

    def bind(self, attention_mask: torch.Tensor) -> "DropoutPlan":
        lens = attention_mask.ne(0).sum(1)
        start = torch.cumsum(lens, 0) - lens
        self.tok = start[:, None] + torch.arange(attention_mask.shape[1])[None, :]
        return self

    def hidden(self, x: torch.Tensor, layer: int, site: int) -> torch.Tensor:
        """x (B, L, d): row = packed token, col = feature."""
        if self.p_hidden <= 0.0:
            return x
        keep = dropout_keep(dropout_site_key(self.seed, self.step, layer, site), self.tok[:, :, None], torch.arange(x.shape[-1])[None, None, :], self.p_hidden)
        return x * (keep.to(x.dtype) * dropout_scale(self.p_hidden))

    def probs(self, pr: torch.Tensor, layer: int) -> torch.Tensor:
        """pr (B, H, L, L): row = packed query token * H + head, col = key position."""
        if self.p_attn <= 0.0:
            return pr
        B, H, L, _ = pr.shape
        rows = self.tok[:, None, :, None] * H + torch.arange(H)[None, :, None, None]
        keep = dropout_keep(dropout_site_key(self.seed, self.step, layer, DROP_SITE_ATTN_PROBS), rows, torch.arange(L)[None, None, None, :], self.p_attn)
        return pr * (keep.to(pr.dtype) * dropout_scale(self.p_attn))

    def global_probs(self, pr: torch.Tensor, layer: int) -> torch.Tensor:
        """pr (B, H, L, L) of the Longformer global projections (only the global query rows are used): row = sequence * H + head."""
        if self.p_attn <= 0.0:
            return pr
        B, H, L, _ = pr.shape
        rows = (torch.arange(B)[:, None, None, None] * H + torch.arange(H)[None, :, None, None]).expand(B, H, L, 1)
        keep = dropout_keep(dropout_site_key(self.seed, self.step, layer, DROP_SITE_GLOBAL_ROW), rows, torch.arange(L)[None, None, None, :], self.p_attn)
        return pr * (keep.to(pr.dtype) * dropout_scale(self.p_attn))


@dataclass
class EncoderConfig:
    hidden: int = 768
    heads: int = 12
    layers: int = 12
    intermediate: int = 3072
    vocab: int = 50265
    max_pos: int = 514
    pad_id: int = 1
    ln_eps: float = 1e-5
    # Recformer extras (module/models/encoder/recformer/interface.py:19-25)
    token_type_size: int = 1
    max_item_embeddings: int = 0
    one_sided_window: int = 0  # 0 => full attention (RoBERTa); 32 => Longformer window 64


def position_ids_from_input_ids(input_ids: torch.Tensor, pad_id: int) -> torch.Tensor:
    """recformer/models.py:64-75 (== transformers create_position_ids_from_input_ids):
    ``cumsum(ids != pad) * (ids != pad) + pad``."""
    mask = input_ids.ne(pad_id).int()
    return (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad_id


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def roberta_embeddings(p: StateDict, input_ids: torch.Tensor, cfg: EncoderConfig, prefix: str = "") -> torch.Tensor:
    """transformers RobertaEmbeddings.forward: ``LN((word[ids] + type[0]) + pos[position_ids])``."""
    pos = position_ids_from_input_ids(input_ids, cfg.pad_id)
    e = p[prefix + "embeddings.word_embeddings.weight"][input_ids]
    e = e + p[prefix + "embeddings.token_type_embeddings.weight"][torch.zeros_like(input_ids)]
    e = e + p[prefix + "embeddings.position_embeddings.weight"][pos]
    return _ln(e, p[prefix + "embeddings.LayerNorm.weight"], p[prefix + "embeddings.LayerNorm.bias"], cfg.ln_eps)


def _ffn_block(p: StateDict, lp: str, attn_ctx: torch.Tensor, x: torch.Tensor, cfg: EncoderConfig, drop=None, layer: int = 0) -> torch.Tensor:
    """BertSelfOutput + BertIntermediate + BertOutput (shared by RoBERTa and Longformer layers):
    ``h = LN(x + dropout(ctx W_o^T + b_o))``; ``y = LN(h + dropout(gelu_erf(h W_i^T + b_i) W_o2^T + b_o2))`` (dropout: train() only)."""
    h = F.linear(attn_ctx, p[lp + "attention.output.dense.weight"], p[lp + "attention.output.dense.bias"])
    if drop is not None:
        h = drop.hidden(h, layer, DROP_SITE_ATTN_OUT)
    h = _ln(h + x, p[lp + "attention.output.LayerNorm.weight"], p[lp + "attention.output.LayerNorm.bias"], cfg.ln_eps)
    i = F.gelu(F.linear(h, p[lp + "intermediate.dense.weight"], p[lp + "intermediate.dense.bias"]))
    o = F.linear(i, p[lp + "output.dense.weight"], p[lp + "output.dense.bias"])
    if drop is not None:
        o = drop.hidden(o, layer, DROP_SITE_FFN_OUT)
    return _ln(o + h, p[lp + "output.LayerNorm.weight"], p[lp + "output.LayerNorm.bias"], cfg.ln_eps)


def roberta_layer(p: StateDict, lp: str, x: torch.Tensor, attention_mask: torch.Tensor, cfg: EncoderConfig, drop=None, layer: int = 0) -> torch.Tensor:
    """transformers RobertaLayer: softmax(QK^T/sqrt(dh) + (1-mask)*finfo.min) V, 12 heads."""
    B, L, d = x.shape
    H, dh = cfg.heads, d // cfg.heads
    q = F.linear(x, p[lp + "attention.self.query.weight"], p[lp + "attention.self.query.bias"]).view(B, L, H, dh).transpose(1, 2)
    k = F.linear(x, p[lp + "attention.self.key.weight"], p[lp + "attention.self.key.bias"]).view(B, L, H, dh).transpose(1, 2)
    v = F.linear(x, p[lp + "attention.self.value.weight"], p[lp + "attention.self.value.bias"]).view(B, L, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * (dh**-0.5)
    s = s + ((1.0 - attention_mask.to(s.dtype)) * torch.finfo(s.dtype).min)[:, None, None, :]
    pr = torch.softmax(s, dim=-1)
    if drop is not None:
        pr = drop.probs(pr, layer)
    ctx = (pr @ v).transpose(1, 2).reshape(B, L, d)
    return _ffn_block(p, lp, ctx, x, cfg, drop, layer)


def roberta_encode(
    p: StateDict, input_ids: torch.Tensor, attention_mask: torch.Tensor, cfg: EncoderConfig, prefix: str = "",
    return_hidden: bool = False, dropout: Optional[DropoutPlan] = None, pooling: str = "cls",
):
    """encoder/_base.py:32-49 -- ``self.model(**batch)`` then ``pool``: CLS ``last_hidden_state[:, 0]`` (:44-45) or "mean"
    ``last_hidden_state.mean(dim=1)`` over the padded length, pad positions included (:42-43).  ``dropout``: the train()-mode
    forward (mask = the build's counter-based function, see DropoutPlan)."""
    x = roberta_embeddings(p, input_ids, cfg, prefix)
    drop = dropout.bind(attention_mask) if dropout is not None else None
    if drop is not None:
        x = drop.hidden(x, 0, DROP_SITE_EMBED)
    hidden = [x]
    for l in range(cfg.layers):
        x = roberta_layer(p, f"{prefix}encoder.layer.{l}.", x, attention_mask, cfg, drop, l)
        hidden.append(x)
    if pooling not in ("cls", "mean"):
        raise ValueError(f"Invalid pooling method: {pooling}.")
    cls = x[:, 0, :] if pooling == "cls" else x.mean(dim=1)
    return (cls, hidden) if return_hidden else cls


# --------------------------------------------------------------------------------------------
# a9/a10/a12: Recformer (Longformer encoder + 4-table embeddings)
# --------------------------------------------------------------------------------------------


def recformer_embeddings(
    p: StateDict, input_ids, token_type_ids, item_position_ids, cfg: EncoderConfig, prefix: str = ""
) -> torch.Tensor:
    """recformer/models.py:104-136 -- ``LN(word + pos + token_type + item_pos)`` in that order."""
    pos = position_ids_from_input_ids(input_ids, cfg.pad_id)
    e = p[prefix + "embeddings.word_embeddings.weight"][input_ids]
    e = e + p[prefix + "embeddings.position_embeddings.weight"][pos]
    e = e + p[prefix + "embeddings.token_type_embeddings.weight"][token_type_ids]
    e = e + p[prefix + "embeddings.item_position_embeddings.weight"][item_position_ids]
    return _ln(e, p[prefix + "embeddings.LayerNorm.weight"], p[prefix + "embeddings.LayerNorm.bias"], cfg.ln_eps)


def longformer_layer(p: StateDict, lp: str, x: torch.Tensor, mask012: torch.Tensor, cfg: EncoderConfig, drop=None, layer: int = 0) -> torch.Tensor:
    """transformers LongformerSelfAttention (reached via recformer/models.py:189,340-348), restated
    densely.  mask012: 0 = no attention, 1 = local, 2 = global (recformer/models.py:261-271).

    local query i (mask 1): softmax over {global keys} U {j : |i-j| <= w, mask[j] == 1}, using the
    ``query/key/value`` projections, query pre-scaled by 1/sqrt(dh); global keys are removed from the
    band and enter once through the extra column; masked keys get finfo.min (probability 0).
    global query g (mask 2): full attention over every non-masked key with the ``*_global``
    projections; its row overwrites the local result.  Masked query rows are zero.
    """
    B, L, d = x.shape
    H, dh, w = cfg.heads, d // cfg.heads, cfg.one_sided_window
    sp = lp + "attention.self."
    scale = 1.0 / math.sqrt(dh)

    def proj(name, t):
        return F.linear(t, p[sp + name + ".weight"], p[sp + name + ".bias"]).view(B, L, H, dh).transpose(1, 2)

    q = proj("query", x) * scale
    k = proj("key", x)
    v = proj("value", x)
    is_masked = mask012 == 0
    is_global = mask012 == 2
    idx = torch.arange(L)
    band = (idx[:, None] - idx[None, :]).abs() <= w  # (L, L)
    # allowed[b, i, j]: key j visible to local query i
    key_local = (mask012 == 1)[:, None, :] & band[None, :, :]
    allowed = key_local | is_global[:, None, :]
    s = q @ k.transpose(-1, -2)  # (B,H,L,L)
    s = s.masked_fill(~allowed[:, None, :, :], float("-inf"))
    pr = torch.softmax(s.to(torch.promote_types(s.dtype, torch.float32)), dim=-1)  # HF: softmax in (at least) fp32; float64 inputs (truth runs) stay float64
    pr = torch.nan_to_num(pr, nan=0.0)
    pr = pr.masked_fill(is_masked[:, None, :, None], 0.0)
    if drop is not None:
        pr = drop.probs(pr, layer)
    ctx = pr @ v  # (B,H,L,dh)
    # global rows
    if is_global.any():
        qg = proj("query_global", x) * scale
        kg = proj("key_global", x)
        vg = proj("value_global", x)
        sg = qg @ kg.transpose(-1, -2)
        sg = sg.masked_fill(is_masked[:, None, None, :], torch.finfo(sg.dtype).min)
        pg = torch.softmax(sg.to(torch.promote_types(sg.dtype, torch.float32)), dim=-1)
        if drop is not None:
            pg = drop.global_probs(pg, layer)
        cg = pg @ vg
        ctx = torch.where(is_global[:, None, :, None], cg, ctx)
    ctx = ctx.transpose(1, 2).reshape(B, L, d)
    return _ffn_block(p, lp, ctx, x, cfg, drop, layer)


def recformer_encode(
    p: StateDict, input_ids, attention_mask, global_attention_mask, token_type_ids, item_position_ids,
    cfg: EncoderConfig, prefix: str = "", return_hidden: bool = False, dropout: Optional[DropoutPlan] = None,
):
    """recformer/interface.py:67-84 -> recformer/models.py:273-361.  Window padding
    (models.py:209-259) only appends masked positions, which never influence a non-masked row in
    the dense restatement, so it is not materialised; CLS pooling per encoder/_base.py:44-45."""
    mask012 = attention_mask * (global_attention_mask + 1)  # models.py:261-271
    x = recformer_embeddings(p, input_ids, token_type_ids, item_position_ids, cfg, prefix)
    drop = dropout.bind(attention_mask) if dropout is not None else None
    if drop is not None:
        x = drop.hidden(x, 0, DROP_SITE_EMBED)
    hidden = [x]
    for l in range(cfg.layers):
        x = longformer_layer(p, f"{prefix}encoder.layer.{l}.", x, mask012, cfg, drop, l)
        hidden.append(x)
    cls = x[:, 0, :]
    return (cls, hidden) if return_hidden else cls


# --------------------------------------------------------------------------------------------
# a14/a16/a17: normalise, score, loss
# --------------------------------------------------------------------------------------------


def maybe_normalize(x: torch.Tensor, similarity: str = "cosine") -> torch.Tensor:
    """module/recommender/module.py:74-77 -- ``F.normalize(x, p=2, dim=-1)`` for cosine."""
    return F.normalize(x, p=2, dim=-1) if similarity == "cosine" else x


def score(user: torch.Tensor, items: torch.Tensor) -> torch.Tensor:
    """module/recommender/module.py:137 -- ``scores = user @ item_embeddings.T``."""
    return user @ items.T


def ce_loss(scores: torch.Tensor, labels: torch.Tensor, temperature: float = 0.05) -> float:
    """module/recommender/module.py:356 -- ``cross_entropy(scores / T, labels)``."""
    return F.cross_entropy(scores / temperature, labels).item()


# --------------------------------------------------------------------------------------------
# a18/a19: top-k and metrics
# --------------------------------------------------------------------------------------------


def topk_canonical(scores: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """evaluator/evaluator.py:43 -- ``torch.topk(scores, k, dim=1)``.  torch leaves the order of
    equal scores unspecified; the build fixes it as (score desc, index asc), NaN first (torch.topk
    treats NaN as the largest value).  Stable descending sort gives exactly that order."""
    key = torch.where(torch.isnan(scores), torch.full_like(scores, float("inf")), scores)
    # push NaN strictly above +inf: sort NaN-flag first
    order = torch.sort(key, dim=1, descending=True, stable=True).indices
    nan_rows = torch.isnan(scores).any(dim=1)
    if nan_rows.any():
        for r in torch.nonzero(nan_rows).flatten().tolist():
            row = scores[r]
            isn = torch.isnan(row)
            nan_idx = torch.nonzero(isn).flatten()
            rest = torch.nonzero(~isn).flatten()
            rest = rest[torch.sort(row[rest], descending=True, stable=True).indices]
            order[r] = torch.cat([nan_idx, rest])
    idx = order[:, :k]
    return torch.gather(scores, 1, idx), idx


def recall_at_k(labels: torch.Tensor, pred: torch.Tensor, k: int) -> float:
    """evaluator/metrics.py:38-59."""
    rows = pred[:, :k].tolist()
    hits = [1.0 if t in r else 0.0 for r, t in zip(rows, labels.tolist())]
    return sum(hits) / len(hits) if hits else 0.0


def ndcg_at_k(labels: torch.Tensor, pred: torch.Tensor, k: int) -> float:
    """evaluator/metrics.py:65-88 -- single positive, gain ``1 / float32(log2(rank + 2))``."""
    rows = pred[:, :k].tolist()
    out = []
    for r, t in zip(rows, labels.tolist()):
        out.append(1 / (torch.log2(torch.tensor(r.index(t) + 2)).item()) if t in r else 0.0)
    return sum(out) / len(out) if out else 0.0


def evaluate(scores: torch.Tensor, labels: torch.Tensor, metrics: Sequence[str], ks: Sequence[int], prefix: str = "") -> Dict[str, float]:
    """evaluator/evaluator.py:31-49 -- metric-major, k-minor key order."""
    _, pred = topk_canonical(scores, max(ks))
    out: Dict[str, float] = OrderedDict()
    for m in metrics:
        for k in ks:
            if m == "NDCG":
                out[f"{prefix}NDCG@{k}"] = ndcg_at_k(labels, pred, k)
            elif m == "RECALL":
                out[f"{prefix}Recall@{k}"] = recall_at_k(labels, pred, k)
            else:
                raise KeyError(m)
    return out


def ranks_equal_up_to_ties(scores: torch.Tensor, idx_a: torch.Tensor, idx_b: torch.Tensor, atol: float) -> bool:
    """Checker helper: two ranked index lists agree if, position by position, either the indices
    are equal or the two indices' oracle scores differ by <= atol (a near-tie swapped by fp32
    summation order); and the score sequence along idx_a is non-increasing within atol."""
    sa = torch.gather(scores, 1, idx_a)
    sb = torch.gather(scores, 1, idx_b)
    ok = (idx_a == idx_b) | ((sa - sb).abs() <= atol)
    return bool(ok.all())


def metrics_after_rank_moves(ref_metrics, ref_rank: torch.Tensor, new_rank: torch.Tensor, ks, prefix: str = "test/", tie_users=None):
    """Checker helper: the reference's Recall@k / NDCG@k (evaluator/metrics.py:38-88: per user 1 / log2(rank + 2) resp. 1 if the label's
    0-based rank is < k, averaged) after the labels of some users moved from ``ref_rank`` to ``new_rank`` -- what the metrics MUST be
    when the only differences to the reference are label moves (each verified separately to cross reference near-ties only).
    Returns (values, slack): ``tie_users`` (bool mask) marks users whose label ties EXACTLY with a neighbour in the reference's scores --
    torch.topk orders such ties arbitrarily, so the reference's own position of those labels is known only to +-1; slack[key] bounds
    what that can change."""
    n = ref_rank.numel()
    out, slack = dict(ref_metrics), {}
    r0, r1 = ref_rank.double(), new_rank.double()
    t = torch.zeros(n, dtype=torch.bool) if tie_users is None else tie_users.bool()
    for k in ks:
        g = lambda r: torch.where((r >= 0) & (r < k), 1.0 / torch.log2(r.clamp(min=0) + 2.0), torch.zeros_like(r))
        h = lambda r: ((r >= 0) & (r < k)).double()
        out[f"{prefix}NDCG@{k}"] = ref_metrics[f"{prefix}NDCG@{k}"] + float((g(r1) - g(r0)).sum()) / n
        out[f"{prefix}Recall@{k}"] = ref_metrics[f"{prefix}Recall@{k}"] + float((h(r1) - h(r0)).sum()) / n
        for name, f in (("NDCG", g), ("Recall", h)):
            step = torch.maximum((f(r0) - f(r0 + 1)).abs(), (f(r0) - f(r0 - 1)).abs())
            slack[f"{prefix}{name}@{k}"] = float(step[t].sum()) / n
    return out, slack


# --------------------------------------------------------------------------------------------
# synthetic weights at true architecture dims (no checkpoints are available offline)
# --------------------------------------------------------------------------------------------


def roberta_param_shapes(cfg: EncoderConfig, prefix: str = "model.", pooler: bool = True):
    """Key order of ``BLaIRBase(...).state_dict()`` = ``'model.' + RobertaModel.state_dict()``
    (models/_base.py:56): 5 embedding tensors, 16 per layer, 2 pooler."""
    d, i = cfg.hidden, cfg.intermediate
    sh = OrderedDict()
    e = prefix + "embeddings."
    sh[e + "word_embeddings.weight"] = (cfg.vocab, d)
    sh[e + "position_embeddings.weight"] = (cfg.max_pos, d)
    sh[e + "token_type_embeddings.weight"] = (cfg.token_type_size, d)
    sh[e + "LayerNorm.weight"] = (d,)
    sh[e + "LayerNorm.bias"] = (d,)
    for l in range(cfg.layers):
        lp = f"{prefix}encoder.layer.{l}."
        for n in ("query", "key", "value"):
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
    if pooler:
        sh[prefix + "pooler.dense.weight"] = (d, d)
        sh[prefix + "pooler.dense.bias"] = (d,)
    return sh


def recformer_param_shapes(cfg: EncoderConfig, prefix: str = "model."):
    """Key order of ``RecformerModel.state_dict()`` (recformer/models.py:84-103,177-193): the
    persistent int64 ``embeddings.position_ids`` buffer comes first (a module's own buffers are
    saved before its children), then 4 tables + LN, then per layer q/k/v, q/k/v_global, out, FFN."""
    d, i = cfg.hidden, cfg.intermediate
    sh = OrderedDict()
    e = prefix + "embeddings."
    sh[e + "position_ids"] = (1, cfg.max_pos)
    sh[e + "word_embeddings.weight"] = (cfg.vocab, d)
    sh[e + "position_embeddings.weight"] = (cfg.max_pos, d)
    sh[e + "token_type_embeddings.weight"] = (cfg.token_type_size, d)
    sh[e + "item_position_embeddings.weight"] = (cfg.max_item_embeddings, d)
    sh[e + "LayerNorm.weight"] = (d,)
    sh[e + "LayerNorm.bias"] = (d,)
    for l in range(cfg.layers):
        lp = f"{prefix}encoder.layer.{l}."
        for n in ("query", "key", "value", "query_global", "key_global", "value_global"):
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
    return sh


def random_state_dict(shapes, seed: int, std: float = 0.02) -> StateDict:
    """HF-style init: N(0, std^2) weights, LayerNorm gamma=1 beta=0, zero biases get
    small noise so bias paths are exercised; ``position_ids`` = arange (int64)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for k, shp in shapes.items():
        if k.endswith("position_ids"):
            sd[k] = torch.arange(shp[-1], dtype=torch.int64).expand(shp).clone()
        elif "LayerNorm.weight" in k:
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            sd[k] = std * torch.randn(shp, generator=g)
        else:
            sd[k] = std * torch.randn(shp, generator=g)
    return sd


# Per-layer query / key gains of the trained-like statistics below, CALIBRATED ONCE in the build container (oracle/calibrate_trained_like.py:
# one sequential pass over a 16-sequence batch, each layer's gain chosen so its pre-softmax logits have sigma = 4) and frozen here as
# constants: the weights are a pure function of (shapes, seed, these numbers), never of a forward pass, so they regenerate bit for bit
# on any host.
TRAINED_LIKE_QK_GAIN = {
    "roberta-base": (0.997, 1.021, 1.103, 0.909, 0.706, 0.685, 0.902, 0.742, 0.64, 1.3, 0.886, 0.965),
    "recformer-base": (0.99, 1.315, 0.866, 0.448, 1.127, 0.88, 1.131, 0.864, 0.913, 0.72, 0.7, 0.811),
}


def trained_like_state_dict(shapes, seed: int, cfg: "EncoderConfig", qk_layer_gain: Sequence[float], *, word_std: float = 0.06,
                            qk_gain: float = 3.0, head_sigma: float = 0.35, vo_gain: float = 4.0, ffn_gain: float = 1.5, n_outlier: int = 3,
                            outlier_gain: float = 5.0, massive: float = 20.0, last_damp: float = 0.1) -> StateDict:
    """Weights with the statistics of a TRAINED encoder instead of HF's init (fixture g22; VERDICT r03 'Missing #2'): what a fine-tuned
    checkpoint (merge_test.py:21-34) stresses and N(0, 0.02^2) never does.

    * query / key projections ``qk_gain`` x wider with a log-normal per-head sharpness and the calibrated per-layer gain: pre-softmax
      logits with sigma ~ 4 and a heavy tail (|max| 20-45, kurtosis 3-14) -- peaky softmax rows, the online-softmax rescale path taken,
      product errors amplified by exp();
    * LayerNorm gamma log-normal (sigma 0.3) with ``n_outlier`` persistent outlier dimensions (x ``outlier_gain``, their beta ~ N(0, 1))
      shared by every LayerNorm, like RoBERTa's dimensions 77 / 588; beta ~ N(0, 0.05^2) elsewhere;
    * three 'massive-activation' hidden dimensions fed by the FFN output bias (+-``massive``, log-normal);
    * the columns of W_q / W_k that read those six dimensions shrunk by 2 / outlier_gain (a trained model's logits stay finite there);
    * word-embedding rows with log-normal norms (sigma 0.5); value / attention-output projections ``vo_gain`` x wider so a token's
      context outweighs its residual: CLS vectors become content-dependent and cosines spread over ~0.2-0.96 instead of crowding 1.0;
    * the last layer's two LayerNorms carry the outlier / massive dimensions at gamma x ``last_damp`` with no outlier beta (contrastive
      fine-tuning flattens uninformative directions of the embedding that is scored).
    """
    g = torch.Generator().manual_seed(seed)
    d, H = cfg.hidden, cfg.heads
    dh = d // H
    perm = torch.randperm(d, generator=g)
    out_dims, mass_dims = perm[:n_outlier], perm[n_outlier:n_outlier + 3]
    hot = torch.cat([out_dims, mass_dims])
    last_tag = f"encoder.layer.{cfg.layers - 1}."
    sd = OrderedDict()
    for k, shp in shapes.items():
        last = last_tag in k
        if k.endswith("position_ids"):
            sd[k] = torch.arange(shp[-1], dtype=torch.int64).expand(shp).clone()
        elif "LayerNorm.weight" in k:
            w = torch.exp(0.3 * torch.randn(shp, generator=g))
            spread = torch.exp(0.3 * torch.randn(n_outlier, generator=g))
            if last:
                w[hot] *= last_damp
            else:
                w[out_dims] *= outlier_gain * spread
            sd[k] = w
        elif "LayerNorm.bias" in k:
            b = 0.05 * torch.randn(shp, generator=g)
            ob = torch.randn(n_outlier, generator=g)
            if not last:
                b[out_dims] += ob
            sd[k] = b
        elif "word_embeddings" in k:
            rows = torch.exp(0.5 * torch.randn(shp[0], 1, generator=g))
            sd[k] = word_std * rows * torch.randn(shp, generator=g)
        elif "embeddings.weight" in k:  # position / token-type / item-position tables
            sd[k] = 0.02 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            b = 0.02 * torch.randn(shp, generator=g)
            if ".output.dense.bias" in k and "attention" not in k:
                b[mass_dims] += massive * torch.sign(torch.randn(3, generator=g)) * torch.exp(0.3 * torch.randn(3, generator=g))
            if ".query" in k or ".key" in k:  # covers query_global / key_global
                b = b * qk_layer_gain[int(k.split("encoder.layer.")[1].split(".")[0])]
            sd[k] = b
        else:
            w = 0.02 * torch.randn(shp, generator=g)
            if ".query" in k or ".key" in k:
                layer = int(k.split("encoder.layer.")[1].split(".")[0])
                hg = qk_gain * qk_layer_gain[layer] * torch.exp(head_sigma * torch.randn(H, generator=g))
                w = (w.view(H, dh, d) * hg[:, None, None]).reshape(shp).clone()
                w[:, hot] *= 2.0 / outlier_gain
            elif ".value" in k or "attention.output.dense.weight" in k:
                w = w * vo_gain
            elif "intermediate.dense.weight" in k or "output.dense.weight" in k:
                w = w * ffn_gain
            sd[k] = w
    return sd


def perturbed_state_dict(base: StateDict, seed: int, std: float = 1e-3) -> StateDict:
    """A synthetic 'fine-tuned' checkpoint: theta_i = theta_pre + tau_i, tau ~ N(0, std^2)."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, v in base.items():
        out[k] = v.clone() if not v.is_floating_point() else v + std * torch.randn(v.shape, generator=g)
    return out


# ---------------------------------------------------------------------------------------------
# next-row 2: distillation losses (rec_retrieval/module/recommender/loss_fn.py) and the teacher
# matrix (merge_train.py:116-126).  Every loss is "mean over rows of a per-row value", which is
# what the fused row kernel computes; the per-row forms below follow the reference line by line.
# ---------------------------------------------------------------------------------------------
DISTILL_LOSSES = (
    "CE", "KD", "MSE", "ADAMERGING", "ADAMERGING_KD", "MERGED_PSEUDO_LABEL", "SINGLE_PSEUDO_LABEL",
    "MERGED_PSEUDO_LABEL_KD", "SINGLE_PSEUDO_LABEL_KD", "PAIRWISE", "LISTNET",
)


def distill_kd(z: torch.Tensor, t: torch.Tensor, temperature: float) -> torch.Tensor:
    """loss_fn.py:52-60 (DistillKDLoss): batchmean KL(softmax(t/T) || softmax(z/T)) * T^2."""
    return F.kl_div(F.log_softmax(z / temperature, dim=-1), F.softmax(t / temperature, dim=-1), reduction="batchmean") * (
        temperature * temperature
    )


def distill_entropy(z: torch.Tensor) -> torch.Tensor:
    """loss_fn.py:64-69 (DistillAdaMergingLoss): mean row entropy with log(p + 1e-8)."""
    p = F.softmax(z, dim=-1)
    return (-torch.sum(p * torch.log(p + 1e-8), dim=-1)).mean()


def distill_pairwise(z: torch.Tensor, t: torch.Tensor, margin: float) -> torch.Tensor:
    """loss_fn.py:183-199 (DistillPairwiseLoss): teacher's best = positive, second best = negative."""
    pos = torch.argmax(t, dim=-1)
    masked = t.clone()
    masked.scatter_(1, pos.unsqueeze(1), float("-inf"))
    neg = torch.argmax(masked, dim=-1)
    ps = z.gather(1, pos.unsqueeze(1)).squeeze(1)
    ns = z.gather(1, neg.unsqueeze(1)).squeeze(1)
    return F.relu(margin - (ps - ns)).mean()


def distill_listnet(z: torch.Tensor, t: torch.Tensor, temperature: float) -> torch.Tensor:
    """loss_fn.py:208-215 (DistillListNetLoss)."""
    return -(F.softmax(t / temperature, dim=-1) * F.log_softmax(z / temperature, dim=-1)).sum(dim=-1).mean()


def distill_loss(name: str, z: torch.Tensor, t: torch.Tensor, temperature: float = 0.05, coefficient: float = 1000.0,
                 margin: float = 0.1) -> torch.Tensor:
    """The reference's loss classes by LossType name (loss_fn.py:37-215; factory :217-267)."""
    if name == "CE" or name == "SINGLE_PSEUDO_LABEL":  # :40-44, :135-142 -- teacher argmax is the label
        return F.cross_entropy(z, torch.argmax(t, dim=-1))
    if name == "KD":
        return distill_kd(z, t, temperature)
    if name == "MSE":  # :171-175
        return F.mse_loss(z, t, reduction="mean")
    if name == "ADAMERGING":
        return distill_entropy(z)
    if name == "ADAMERGING_KD":  # :82-88
        return distill_entropy(z) + coefficient * distill_kd(z, t, temperature)
    if name == "MERGED_PSEUDO_LABEL":  # :95-104 -- the student's own argmax is the label
        return F.cross_entropy(z, torch.argmax(z, dim=-1))
    if name == "MERGED_PSEUDO_LABEL_KD":  # :112-125
        return F.cross_entropy(z, torch.argmax(z, dim=-1)) + coefficient * distill_kd(z, t, temperature)
    if name == "SINGLE_PSEUDO_LABEL_KD":  # :150-163 (cfg5's loss: T = 0.05, coefficient = 1000)
        return F.cross_entropy(z, torch.argmax(t, dim=-1)) + coefficient * distill_kd(z, t, temperature)
    if name == "PAIRWISE":
        return distill_pairwise(z, t, margin)
    if name == "LISTNET":
        return distill_listnet(z, t, temperature)
    raise ValueError(name)


def teacher_scores(sequence_embedding: torch.Tensor, item_embedding: torch.Tensor) -> torch.Tensor:
    """merge_train.py:120-126: rows normalised by x / x.norm(dim=-1, keepdim=True), then S = seq @ item.T."""
    item = item_embedding / item_embedding.norm(dim=-1, keepdim=True)
    seq = sequence_embedding / sequence_embedding.norm(dim=-1, keepdim=True)
    return seq @ item.T


def forward_distill(reps: torch.Tensor, item_embeddings, score_embeddings, dataset_indexes, sequence_ids, loss) -> torch.Tensor:
    """distiller/sequence/module.py:59-74: per-sample logits rep_i @ E_ds.T against the teacher row, mean over the batch."""
    losses = []
    for i, (ds, sid) in enumerate(zip(dataset_indexes, sequence_ids)):
        logit = reps[i] @ item_embeddings[ds].T
        losses.append(loss(logit.unsqueeze(0), score_embeddings[ds][sid].unsqueeze(0)))
    return torch.stack(losses).mean()


# --------------------------------------------------------------------------------------------
# fine-tuning (finetune_train.py): negative-sampling scores, loss, optimizer groups, schedule, AdamW
# pinned by tests/golden/g9_finetune.pt (oracle/gen_golden_finetune.py ran the reference's RecModule)
# --------------------------------------------------------------------------------------------
def negative_sample_scores(user: torch.Tensor, target: torch.Tensor, negatives: Optional[torch.Tensor], mode: str,
                           k: Optional[int]) -> Tuple[torch.Tensor, torch.Tensor]:
    """module/recommender/module.py:79-131 on already encoded + normalised rows -> (scores, labels)."""
    B = user.shape[0]
    if mode == "IN_BATCH":  # :92-94
        return user @ target.T, torch.arange(B)
    assert negatives is not None, "negative_batch must not be None"
    neg = negatives.reshape(B, k, -1)
    if mode == "SAMPLE":  # :96-109: [target | own negatives], label 0
        allenc = torch.cat((target.unsqueeze(1), neg), dim=1)
        return torch.bmm(user.unsqueeze(1), allenc.transpose(1, 2)).squeeze(1), torch.zeros(B, dtype=torch.long)
    if mode == "IN_BATCH_SAMPLE":  # :111-126: [every target of the batch | own negatives], label = row
        own = torch.bmm(user.unsqueeze(1), neg.transpose(1, 2)).squeeze(1)
        return torch.cat((user @ target.T, own), dim=1), torch.arange(B)
    raise ValueError(f"Invalid negative sample mode: {mode}")


def finetune_loss(scores: torch.Tensor, labels: torch.Tensor, temperature: float = 0.05) -> torch.Tensor:
    """module.py:183: cross_entropy(scores / temperature, labels)."""
    return F.cross_entropy(scores / temperature, labels)


def optimizer_groups(names: Sequence[str], weight_decay: float) -> "OrderedDict[str, float]":
    """module.py:45-56: weight decay per parameter name -- 0 for names containing "bias" or "LayerNorm.weight"."""
    no_decay = ("bias", "LayerNorm.weight")
    return OrderedDict((n, 0.0 if any(nd in n for nd in no_decay) else weight_decay) for n in names)


def resolve_warmup(warmup_steps, estimated_stepping_batches: int):
    """module.py:58-63: a float is a fraction of all optimizer steps, an int a step count."""
    if isinstance(warmup_steps, float):
        return estimated_stepping_batches * warmup_steps
    if isinstance(warmup_steps, int):
        return warmup_steps
    raise ValueError(f"Invalid warmup_steps type {type(warmup_steps)}")


def linear_warmup_multiplier(step: int, num_warmup_steps, num_training_steps) -> float:
    """transformers.get_linear_schedule_with_warmup (module.py:66-70), the lambda of its LambdaLR: the multiplier in force for the
    optimizer step taken after ``step`` earlier ones."""
    if step < num_warmup_steps:
        return float(step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))


def clip_coefficient(grads: Sequence[torch.Tensor], max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (Lightning's gradient_clip_val, finetune_train.py:106): min(1, max_norm / (||g||_2 + 1e-6))."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads]))
    return torch.clamp(max_norm / (total + 1e-6), max=1.0)


def adamw_step(p: torch.Tensor, g: torch.Tensor, m: torch.Tensor, v: torch.Tensor, lr: float, weight_decay: float, step: int,
               betas=(0.9, 0.999), eps: float = 1e-8) -> None:
    """torch.optim.AdamW (module.py:65), single-tensor update, in place on p, m, v; ``step`` counts from 1."""
    b1, b2 = betas
    p.mul_(1 - lr * weight_decay)
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
