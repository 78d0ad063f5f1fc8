"""Thin torch-tensor wrappers over the C ABI (include/mergerec_hip.h).

Every function launches hand-written HIP kernels on the current torch stream and returns torch tensors
that own the output memory.  No function here has a torch/eager fallback: inputs must be CUDA(=HIP)
tensors and the shared library must be present, otherwise MergeRecHipError / ValueError is raised.
"""
from __future__ import annotations

import functools

import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import MergeRecHipError, check, ptr

ACT_NONE, ACT_GELU, ACT_TANH = 0, 1, 2
EMBED_ROBERTA, EMBED_RECFORMER = 0, 1


class LaunchProfiler:
    """Optional per-launch timing with HIP events recorded on the stream the kernel is launched on
    (torch's current stream).  bench.py turns it on for the timed region; records are
    (kernel family, algorithmic flops, algorithmic bytes, start event, end event)."""

    def __init__(self):
        self.enabled = False
        self.records = []
        self._tail = None     # end event of the last profiled launch
        self._chain = False   # set by chain(): the next begin() reuses that event instead of recording another one

    def chain(self):
        """Declare that NOTHING was enqueued on the stream since the last profiled launch ended: the next launch's start stamp is that
        launch's end stamp (one event record between two back-to-back kernels instead of two -- an event record costs the stream about
        2.5 us, 0.5 ms per bench step at two per launch).  The callers are the encoder's layer body and the staged scoring pair."""
        self._chain = self.enabled

    def begin(self, dev):
        if not self.enabled:
            return None
        if self._chain and self._tail is not None:
            self._chain = False
            return self._tail
        self._chain = False
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(dev))
        return ev

    def end(self, ev0, dev, name, flops=0.0, nbytes=0.0):
        if ev0 is None:
            return
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(torch.cuda.current_stream(dev))
        self._tail = ev1
        self.records.append((name, float(flops), float(nbytes), ev0, ev1))

    def summary(self):
        """name -> dict(launches, ms, flops, bytes); call after a device synchronize."""
        out = {}
        for name, fl, nb, e0, e1 in self.records:
            d = out.setdefault(name, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += nb
        return out


PROF = LaunchProfiler()


def _dev(t: torch.Tensor, name: str, dtype=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise ValueError(f"{name} must be a GPU tensor (the HIP path has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def h2d(t: torch.Tensor, device) -> torch.Tensor:
    """A small host tensor to the device WITHOUT stalling the host: staged through pinned memory (torch's caching host allocator, which
    keeps the block until the copy has run), so the copy is one more stream-ordered command.  From pageable memory the same call blocks
    the host until everything queued before it has finished -- once per batch that drains the launch queue the host had built up."""
    if t.is_cuda:
        return t.to(device)
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    return p.to(device, non_blocking=True)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # the handle without building a torch.cuda.Stream object per launch


def _stream(t: torch.Tensor):
    """HIP stream handle of torch's current stream on ``t``'s device (every kernel is launched on it)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


# ------------------------------------------------------------------------------------------ merger
def task_vector(theta: torch.Tensor, base: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(theta, "theta", torch.float32), _dev(base, "base", torch.float32)
    if theta.numel() != base.numel():
        raise ValueError("theta/base size mismatch")
    out = torch.empty_like(base) if out is None else _dev(out, "out", torch.float32)
    check(_lib.load().mr_task_vector_f32(ptr(theta), ptr(base), base.numel(), ptr(out), _stream(base)), "mr_task_vector_f32")
    return out


def merge_nway(
    base: torch.Tensor, tv: torch.Tensor, alpha: torch.Tensor, seg_off: Optional[torch.Tensor] = None,
    out: Optional[torch.Tensor] = None, p_begin: int = 0, p_count: Optional[int] = None, out_is_slice: bool = False,
    operands_are_slices: bool = False,
) -> torch.Tensor:
    """out[p] = base[p] + sum_i alpha[s(p), i] * tv[i, p] for p in [p_begin, p_begin + p_count).
    out_is_slice: `out` holds only the slice (p_count floats; out[0] is element p_begin) -- used when a rank
    merges its arena slice into a send buffer for the all-gather.
    operands_are_slices: `base` / `tv` hold only that slice too (base[0], tv[i, 0] are element p_begin): the rank never stored the rest."""
    _dev(base, "base", torch.float32), _dev(tv, "tv", torch.float32), _dev(alpha, "alpha", torch.float32)
    if tv.dim() != 2 or tv.shape[1] != base.numel():
        raise ValueError("tv must be (N, P)")
    N, P = tv.shape
    S = 1 if seg_off is None else seg_off.numel() - 1
    if alpha.numel() != S * N:
        raise ValueError(f"alpha must hold S*N = {S * N} floats, got {alpha.numel()}")
    if seg_off is not None:
        _dev(seg_off, "seg_off", torch.int64)
    out = torch.empty_like(base) if out is None else _dev(out, "out", torch.float32)
    if operands_are_slices:
        if p_count is None:
            p_count = P
        if p_begin < 0 or p_begin % 4 or p_count < 0 or p_count > P:
            raise ValueError("slice out of range")
    else:
        p_count = P - p_begin if p_count is None else p_count
        if p_begin < 0 or p_count < 0 or p_begin + p_count > P:
            raise ValueError("slice out of range")
    if out.numel() < (p_count if out_is_slice else p_begin + p_count):
        raise ValueError("out is too small for the requested slice")
    out_ptr = out.data_ptr() - (4 * p_begin if out_is_slice else 0)
    shift = 4 * p_begin if operands_are_slices else 0  # the kernel indexes with global p: rebase the slice-only operands
    ev = PROF.begin(base.device)
    check(
        _lib.load().mr_merge_nway_f32(base.data_ptr() - shift, tv.data_ptr() - shift, tv.stride(0), ptr(alpha), ptr(seg_off), N, S, p_begin, p_count,
                                      out_ptr, _stream(base)),
        "mr_merge_nway_f32",
    )
    PROF.end(ev, base.device, "merge_nway", nbytes=(N + 2) * p_count * 4)
    return out


def merge_rows(base: torch.Tensor, tv: torch.Tensor, alpha_row: torch.Tensor, idx: torch.Tensor, rows: int, d: int, table_off: int,
               out: torch.Tensor) -> torch.Tensor:
    """out[table_off + r d + c] = base[..] + sum_i alpha_row[i] tv[i, ..] for the rows r = idx[t] of the (rows, d) table at arena offset
    ``table_off`` -- the same operations per element as ``merge_nway`` (bit-identical rows); the rest of ``out`` is not touched."""
    _dev(base, "base", torch.float32), _dev(tv, "tv", torch.float32), _dev(alpha_row, "alpha_row", torch.float32), _dev(idx, "idx", torch.int32)
    _dev(out, "out", torch.float32)
    N, P = tv.shape
    if alpha_row.numel() != N or table_off < 0 or table_off + rows * d > min(P, base.numel(), out.numel()):
        raise ValueError("alpha_row must hold N coefficients and the table must lie inside the vectors")
    check(_lib.load().mr_merge_rows_f32(ptr(base), ptr(tv), tv.stride(0), ptr(alpha_row), N, ptr(idx), idx.numel(), rows, d, table_off, ptr(out),
                                        _stream(base)), "mr_merge_rows_f32")
    return out


def merge_running(base: Optional[torch.Tensor], models: torch.Tensor, weights: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ModelMerger.merge's running sums: base given -> "task_vector" (base + w_0 (m_0 - base) + ...), base None -> "linear"."""
    _dev(models, "models", torch.float32), _dev(weights, "weights", torch.float32)
    if models.dim() != 2 or weights.numel() != models.shape[0]:
        raise ValueError("models must be (N, P) with one weight per row")
    N, P = models.shape
    if base is not None:
        _dev(base, "base", torch.float32)
        if base.numel() != P:
            raise ValueError("base / models size mismatch")
    out = torch.empty(P, dtype=torch.float32, device=models.device) if out is None else _dev(out, "out", torch.float32)
    ev = PROF.begin(models.device)
    check(_lib.load().mr_merge_running_f32(ptr(base), ptr(models), models.stride(0), ptr(weights), N, P, ptr(out), _stream(models)), "mr_merge_running_f32")
    PROF.end(ev, models.device, "merge_running", nbytes=(N + (2 if base is not None else 1)) * P * 4)
    return out


def merge_bwd_alpha(tv: torch.Tensor, g: torch.Tensor, seg_off: Optional[torch.Tensor] = None, p_begin: int = 0,
                    p_count: Optional[int] = None) -> torch.Tensor:
    """dalpha[s, i] = <tv[i, segment s], g[segment s]>.  ``p_begin`` / ``p_count`` (multiples of 4; one segment only): the contraction over
    the arena range [p_begin, p_begin + p_count) alone -- the alpha-learning step contracts each layer's range as soon as its gradients exist."""
    _dev(tv, "tv", torch.float32), _dev(g, "g", torch.float32)
    N, P = tv.shape
    S = 1 if seg_off is None else seg_off.numel() - 1
    shift = 0
    if p_begin or p_count is not None:
        p_count = P - p_begin if p_count is None else p_count
        if seg_off is not None or p_begin < 0 or p_count < 0 or p_begin + p_count > P or (p_begin | p_count) & 3 or g.numel() < p_begin + p_count:
            raise ValueError("a sub-range contraction takes one segment, a range inside the vectors, offsets in multiples of 4")
        shift, P = 4 * p_begin, p_count
    lib = _lib.load()
    nbytes = lib.mr_merge_bwd_alpha_ws_bytes(N, S, P)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=tv.device)
    out = torch.empty(S, N, dtype=torch.float32, device=tv.device)
    check(lib.mr_merge_bwd_alpha_f32(tv.data_ptr() + shift, tv.stride(0), g.data_ptr() + shift, ptr(seg_off), N, S, P, ptr(out), ptr(ws), nbytes,
                                     _stream(tv)), "mr_merge_bwd_alpha_f32")
    return out


def abs_topk_mask(x: torch.Tensor, k: int, out: Optional[torch.Tensor] = None, want_mask: bool = False):
    """y = x where |x| is among the k largest of the vector (ties at the threshold -> lowest indices), else 0.
    Returns (y, mask uint8 or None).  Exact radix select + ordered tie ranking on the device, no host sync."""
    _dev(x, "x", torch.float32)
    n = x.numel()
    y = torch.empty_like(x) if out is None else out
    m = torch.empty(n, dtype=torch.uint8, device=x.device) if want_mask else None
    if k <= 0:
        y.zero_()
        if m is not None:
            m.zero_()
        return y, m
    lib = _lib.load()
    nbytes = lib.mr_select_ws_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    thr = torch.empty(1, dtype=torch.int32, device=x.device)
    need = torch.empty(1, dtype=torch.int64, device=x.device)
    check(lib.mr_abs_kth_largest_f32(ptr(x), n, min(k, n), ptr(thr), ptr(need), ptr(ws), nbytes, _stream(x)), "mr_abs_kth_largest_f32")
    check(lib.mr_abs_topk_mask_f32(ptr(x), n, ptr(thr), ptr(need), ptr(y), ptr(m), ptr(ws), nbytes, _stream(x)), "mr_abs_topk_mask_f32")
    return y, m


def kth_largest_value(x: torch.Tensor, k: int, signed: bool, out: torch.Tensor) -> torch.Tensor:
    """out[0] (device float) = k-th largest element of x by value (signed) or by magnitude."""
    _dev(x, "x", torch.float32)
    lib = _lib.load()
    n = x.numel()
    nbytes = lib.mr_select_ws_bytes(n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    check(lib.mr_kth_largest_value_f32(ptr(x), n, k, int(signed), ptr(out), ptr(ws), nbytes, _stream(x)), "mr_kth_largest_value_f32")
    return out


def pcb_vectors(tv: torch.Tensor, density: float) -> torch.Tensor:
    """merger/algorithms/pcb.py:37-58 on a COMPACT (N, P) task-vector matrix; returns the (N, P) PCB vectors."""
    _dev(tv, "tv", torch.float32)
    N, d = tv.shape
    lib = _lib.load()
    q1 = torch.empty(N, 2, dtype=torch.float32, device=tv.device)
    q2 = torch.empty(N, 2, dtype=torch.float32, device=tv.device)
    clamped, task = torch.empty_like(tv), torch.empty_like(tv)
    j_lo, j_hi = int(d * 0.01), int(d * (1 - 0.01) - 1)          # _clamp(|tv|, 0.01, 0.01): ascending indices
    for i in range(N):
        kth_largest_value(tv[i], d - j_lo, False, q1[i, 0:1])
        kth_largest_value(tv[i], d - j_hi, False, q1[i, 1:2])
        check(lib.mr_pcb_stage1_f32(ptr(tv), tv.stride(0), N, i, d, ptr(q1[i]), ptr(clamped[i]), ptr(task[i]), _stream(tv)), "mr_pcb_stage1_f32")
    j_lo2, j_hi2 = int(d * (1 - density)), int(d * (1 - 0) - 1)  # _clamp(task_pcb, 1 - density, 0)
    for i in range(N):
        kth_largest_value(task[i], d - j_lo2, True, q2[i, 0:1])
        kth_largest_value(task[i], d - j_hi2, True, q2[i, 1:2])
    out = torch.empty_like(tv)
    check(lib.mr_pcb_stage2_f32(ptr(clamped), ptr(task), tv.stride(0), N, d, ptr(q2), ptr(out), _stream(tv)), "mr_pcb_stage2_f32")
    return out


def ties_combine(sparse: torch.Tensor) -> torch.Tensor:
    """In place on (N, P) masked updates: TIES sign election + disjoint mean."""
    _dev(sparse, "sparse", torch.float32)
    N, P = sparse.shape
    check(_lib.load().mr_ties_combine_f32(ptr(sparse), sparse.stride(0), N, P, _stream(sparse)), "mr_ties_combine_f32")
    return sparse


def lns_combine(tv: torch.Tensor, mask: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(tv, "tv", torch.float32), _dev(mask, "mask", torch.uint8)
    N, P = tv.shape
    out = torch.empty_like(tv) if out is None else out
    check(_lib.load().mr_lns_combine_f32(ptr(tv), ptr(mask), tv.stride(0), N, P, ptr(out), _stream(tv)), "mr_lns_combine_f32")
    return out


# ------------------------------------------------------------------------------------------ encoder
INPUT_ERRORS = {1: "input_ids outside [0, vocab)", 2: "position 0 (CLS) must be attended: CLS pooling reads it (encoder/_base.py:45)",
                4: "token_type_ids out of range", 8: "item_position_ids out of range",
                16: "only the Recformer pattern (global attention on token 0 only) is built",
                32: "attention_mask row counts disagree with the lengths handed to the packer"}


def pack_tokens(input_ids, attention_mask, cu_seqlens, T: int, pad_id: int, token_type_ids=None, item_position_ids=None,
                global_attention_mask=None, err_bits: Optional[torch.Tensor] = None, vocab: int = 0, n_type: int = 0, n_ip: int = 0):
    """Padded (B, L) id tensors -> packed per-token index arrays.  ``err_bits`` (int32 device word, zeroed by the caller): the kernel
    ORs INPUT_ERRORS bits into it instead of the host validating with reductions + a device sync per batch."""
    _dev(input_ids, "input_ids", torch.int64), _dev(attention_mask, "attention_mask", torch.int64)
    _dev(cu_seqlens, "cu_seqlens", torch.int32)
    B, L = input_ids.shape
    dev = input_ids.device
    tok_word = torch.empty(T, dtype=torch.int32, device=dev)
    tok_pos = torch.empty(T, dtype=torch.int32, device=dev)
    tok_tt = torch.empty(T, dtype=torch.int32, device=dev) if token_type_ids is not None else None
    tok_ip = torch.empty(T, dtype=torch.int32, device=dev) if item_position_ids is not None else None
    for t, n in ((token_type_ids, "token_type_ids"), (item_position_ids, "item_position_ids"), (global_attention_mask, "global_attention_mask")):
        if t is not None:
            _dev(t, n, torch.int64)
            if t.shape != input_ids.shape:
                raise ValueError(f"{n} must have the shape of input_ids")
    if err_bits is not None:
        _dev(err_bits, "err_bits", torch.int32)
    check(
        _lib.load().mr_pack_tokens_checked(ptr(input_ids), ptr(attention_mask), ptr(token_type_ids), ptr(item_position_ids),
                                           ptr(global_attention_mask if err_bits is not None else None), B, L, pad_id, vocab, n_type, n_ip,
                                           ptr(cu_seqlens), ptr(tok_word), ptr(tok_pos), ptr(tok_tt), ptr(tok_ip), ptr(err_bits),
                                           _stream(input_ids)),
        "mr_pack_tokens_checked",
    )
    return tok_word, tok_pos, tok_tt, tok_ip


def embed_gather_ln(tok_word, tok_pos, tok_tt, tok_ip, word, pos, type_, itempos, gamma, beta, eps: float, mode: int, out=None):
    T, d = tok_word.numel(), word.shape[1]
    out = torch.empty(T, d, dtype=torch.float32, device=word.device) if out is None else out
    ev = PROF.begin(word.device)
    check(
        _lib.load().mr_embed_gather_ln_f32(
            ptr(tok_word), ptr(tok_pos), ptr(tok_tt), ptr(tok_ip), ptr(word), ptr(pos), ptr(type_), ptr(itempos),
            word.shape[0], pos.shape[0], type_.shape[0], 0 if itempos is None else itempos.shape[0],
            ptr(gamma), ptr(beta), eps, T, d, mode, ptr(out), _stream(word)),
        "mr_embed_gather_ln_f32",
    )
    n_id = 2 + (tok_tt is not None) + (tok_ip is not None)
    PROF.end(ev, word.device, "embed_gather_ln", nbytes=T * (2 * d * 4 + 4 * n_id))
    return out


def gemm_nt(A: torch.Tensor, weights: Sequence[torch.Tensor], biases: Sequence[Optional[torch.Tensor]] = (None,),
            act: int = ACT_NONE, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
            prof_name: str = "gemm_nt") -> torch.Tensor:
    """out[:, s*n:(s+1)*n] = act(A @ weights[s].T + biases[s]) (+ residual); 1..3 equally-shaped weight segments."""
    if A.dim() != 2 or A.stride(1) != 1:
        raise ValueError("A must be 2-D with unit inner stride")
    nseg = len(weights)
    seg_n, K = weights[0].shape
    if A.shape[1] != K:
        raise ValueError("K mismatch")
    biases = list(biases) + [None] * (3 - len(biases))
    ws = list(weights) + [None] * (3 - nseg)
    M = A.shape[0]
    out = torch.empty(M, nseg * seg_n, dtype=torch.float32, device=A.device) if out is None else out
    ev = PROF.begin(A.device)
    check(
        _lib.load().mr_gemm_nt_bias_act_f32(
            ptr(A), A.stride(0), ptr(ws[0]), ptr(ws[1]), ptr(ws[2]), ptr(biases[0]), ptr(biases[1]), ptr(biases[2]), nseg, M, seg_n, K,
            act, ptr(residual), 0 if residual is None else residual.stride(0), ptr(out), out.stride(0), _stream(A)),
        "mr_gemm_nt_bias_act_f32",
    )
    PROF.end(ev, A.device, prof_name, flops=2.0 * M * nseg * seg_n * K, nbytes=4.0 * (M * K + nseg * seg_n * K + M * nseg * seg_n * (2 if residual is not None else 1)))
    return out


def split_bf16x3(x: torch.Tensor, pieces=None):
    """x (fp32, numel % 4 == 0) -> (hi, mid, lo) bf16 tensors of the same length (mr_split_bf16x3_f32)."""
    _dev(x, "x", torch.float32)
    n = x.numel()
    if pieces is None:
        pieces = tuple(torch.empty(n, dtype=torch.bfloat16, device=x.device) for _ in range(3))
    ev = PROF.begin(x.device)
    check(_lib.load().mr_split_bf16x3_f32(ptr(x), n, ptr(pieces[0]), ptr(pieces[1]), ptr(pieces[2]), _stream(x)), "mr_split_bf16x3_f32")
    PROF.end(ev, x.device, "split_bf16x3", nbytes=n * 10.0)
    return pieces


class KBlockTable:
    """Device table of the weight matrices to pre-split: rows (arena offset, N, K)."""

    def __init__(self, entries, device):
        ent = [(int(o), int(n), int(k)) for o, n, k in entries]
        for o, n, k in ent:
            if k % 16 or o % 8:
                raise ValueError("k-blocked split needs K % 16 == 0 and offset % 8 == 0")
        self.n_mat = len(ent)
        units = [n * k // 4 for _, n, k in ent]
        pref = [0]
        for u in units:
            pref.append(pref[-1] + u)
        self.total_units = pref[-1]
        self.table = torch.tensor([x for e in ent for x in e], dtype=torch.int64, device=device)
        self.prefix = torch.tensor(pref, dtype=torch.int64, device=device)
        self.bytes = sum(n * k for _, n, k in ent) * 10.0  # 4 B read + 3 x 2 B written per element


PRODUCTS_F16X3 = 35  # include/mergerec_hip.h MR_PRODUCTS_F16X3: two fp16 pieces per operand, three products


def split_weights_kblock(flat: torch.Tensor, table: KBlockTable, pieces=None, n_pieces: int = 3, f16: bool = False, overflow=None):
    """fp32 arena -> bf16 piece arenas (same length) holding the listed matrices in k-blocked form.  n_pieces = 2: only (hi, mid)
    are produced -- all that the three-product GEMMs read -- and the returned tuple's third entry is None.  f16: the two FP16 pieces of
    256 w for the "f16x3" arithmetic (``overflow``: device int32 flag the kernel raises when a scaled weight leaves fp16's range)."""
    _dev(flat, "flat", torch.float32)
    dt = torch.float16 if f16 else torch.bfloat16
    if f16:
        n_pieces = 2
    if pieces is None or sum(p is not None for p in pieces) != n_pieces or pieces[0].dtype != dt:
        pieces = tuple(torch.zeros(flat.numel(), dtype=dt, device=flat.device) if i < n_pieces else None for i in range(3))
    ev = PROF.begin(flat.device)
    if f16:
        check(_lib.load().mr_split_weights_kblock_f16_f32(ptr(flat), ptr(table.table), ptr(table.prefix), table.n_mat, table.total_units,
                                                          ptr(pieces[0]), ptr(pieces[1]), ptr(overflow), _stream(flat)), "mr_split_weights_kblock_f16_f32")
    else:
        check(_lib.load().mr_split_weights_kblock_f32(ptr(flat), ptr(table.table), ptr(table.prefix), table.n_mat, table.total_units,
                                                      ptr(pieces[0]), ptr(pieces[1]), ptr(pieces[2]), _stream(flat)), "mr_split_weights_kblock_f32")
    PROF.end(ev, flat.device, "split_weights", nbytes=table.bytes * (4 + 2 * n_pieces) / 10.0)
    return pieces


def gemm_nt_split(A: torch.Tensor, pieces, offsets: Sequence[int], seg_n: int, K: int, biases: Sequence[Optional[torch.Tensor]] = (None,),
                  act: int = ACT_NONE, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                  products: int = 6) -> torch.Tensor:
    """Split-precision GEMM (6 bf16 MFMA products per fp32 product): weights are addressed as element offsets into the
    three pre-split, K-BLOCKED bf16 arenas `pieces` = (hi, mid, lo) (split_weights_kblock); otherwise as gemm_nt."""
    if A.dim() != 2 or A.stride(1) != 1 or A.shape[1] != K:
        raise ValueError("A must be (M, K) with unit inner stride")
    nseg = len(offsets)
    offs = list(offsets) + [0] * (3 - nseg)
    biases = list(biases) + [None] * (3 - len(biases))
    M = A.shape[0]
    out = torch.empty(M, nseg * seg_n, dtype=torch.float32, device=A.device) if out is None else out
    ev = PROF.begin(A.device)
    check(
        _lib.load().mr_gemm_nt_bf16x6_f32(
            ptr(A), A.stride(0), ptr(pieces[0]), ptr(pieces[1]), ptr(pieces[2]), offs[0], offs[1], offs[2], ptr(biases[0]), ptr(biases[1]),
            ptr(biases[2]), nseg, M, seg_n, K, act, ptr(residual), 0 if residual is None else residual.stride(0), ptr(out), out.stride(0),
            products, _stream(A)),
        "mr_gemm_nt_bf16x6_f32",
    )
    PROF.end(ev, A.device, {6: "gemm_nt_bf16x6", 3: "gemm_nt_bf16x3", PRODUCTS_F16X3: "gemm_nt_f16x3"}[products], flops=2.0 * M * nseg * seg_n * K,
             nbytes=4.0 * (M * K + M * nseg * seg_n * (2 if residual is not None else 1)) + 6.0 * nseg * seg_n * K)
    return out


def layernorm(x: torch.Tensor, gamma, beta, eps: float, out=None) -> torch.Tensor:
    T, d = x.shape
    out = torch.empty(T, d, dtype=torch.float32, device=x.device) if out is None else out
    ev = PROF.begin(x.device)
    check(_lib.load().mr_layernorm_f32(ptr(x), x.stride(0), ptr(gamma), ptr(beta), eps, T, d, ptr(out), out.stride(0), _stream(x)), "mr_layernorm_f32")
    PROF.end(ev, x.device, "layernorm", nbytes=2.0 * T * d * 4)
    return out


ATTN_FLOPS_HINT = [0.0]  # algorithmic 4 * sum(L_b^2) * d of the next attention launch (set by the engine)


def attn_work_plan(lens: torch.Tensor, q_rows: int) -> Tuple[torch.Tensor, int]:
    """Host-side work list of the split attention kernels (mr_attn_work_plan): lens = int64 CPU tensor (B) -> (int32 CPU tensor
    (n_slots * 8), n_slots).  Pure host code: no GPU call, no synchronisation."""
    lens = lens.to(torch.int64).contiguous()
    lib = _lib.load()
    n = int(lib.mr_attn_work_plan(ptr(lens), lens.numel(), q_rows, None, 0))
    if n < 0:
        check(n, "mr_attn_work_plan")
    work = torch.empty(n * 8, dtype=torch.int32)
    if n:
        got = int(lib.mr_attn_work_plan(ptr(lens), lens.numel(), q_rows, ptr(work), work.numel()))
        assert got == n
    return work, n


def attn_q_rows(window: int, products: int) -> int:
    return int(_lib.load().mr_attn_split_q_rows(window, products))


def attention(qkv: torch.Tensor, cu_seqlens: torch.Tensor, B: int, H: int, max_len: int, window: int = -1, out=None,
              seq_order: Optional[torch.Tensor] = None, products: int = 0, work=None, drop_p: float = 0.0, drop_key: int = 0) -> torch.Tensor:
    """products = 0: exact fp32 MFMA kernel; 3 / 6: split-precision bf16 MFMA kernel.  work: {q_rows: (device int32 work list, n_slots)}
    from ``attn_work_plan`` (the engine builds it while packing) -> the work-list launch (mr_attn_split_work_f32); without it the
    (max_len / 128, H, B) grid of mr_attn_split_f32.  drop_p > 0 (training graph only): dropout on the attention probabilities with the
    mask of csrc/dropout.h under ``drop_key``."""
    T = qkv.shape[0]
    dh = qkv.shape[1] // (3 * H)
    out = torch.empty(T, H * dh, dtype=torch.float32, device=qkv.device) if out is None else out
    ev = PROF.begin(qkv.device)
    use_list = bool(work) and os.environ.get("MR_ATTN_WORKLIST", "1") != "0"
    lib = _lib.load()
    if products and use_list and (drop_p == 0.0 or products == 3) and work.get(attn_q_rows(window, products)) is not None:
        qr = attn_q_rows(window, products)  # the list's key IS the block height it was planned with; the entry point checks it (and B)
        wl = work[qr]
        if drop_p > 0.0:
            check(lib.mr_attn_split_work_train_f32(ptr(qkv), ptr(cu_seqlens), ptr(wl[0]), wl[1], B, qr, H, dh, dh ** -0.5, window, products, drop_p, drop_key,
                                                   ptr(out), _stream(qkv)), "mr_attn_split_work_train_f32")
        else:
            check(lib.mr_attn_split_work_f32(ptr(qkv), ptr(cu_seqlens), ptr(wl[0]), wl[1], B, qr, H, dh, dh ** -0.5, window, products, ptr(out), _stream(qkv)),
                  "mr_attn_split_work_f32")
    elif (not products or drop_p > 0.0) and use_list and work.get(128) is not None:
        # exact-fp32 kernel on the work list (also the route of a dropout launch in the six-product mode, which has no dropout variant)
        wl = work[128]
        check(lib.mr_attn_work_f32(ptr(qkv), ptr(cu_seqlens), ptr(wl[0]), wl[1], B, 128, H, dh, dh ** -0.5, window, float(drop_p), drop_key, ptr(out), _stream(qkv)),
              "mr_attn_work_f32")
    elif drop_p > 0.0:
        check(lib.mr_attn_train_f32(ptr(qkv), ptr(cu_seqlens), ptr(seq_order), B, H, dh, max_len, dh ** -0.5, window, drop_p, drop_key, ptr(out), _stream(qkv)),
              "mr_attn_train_f32")
    elif products:
        check(lib.mr_attn_split_f32(ptr(qkv), ptr(cu_seqlens), ptr(seq_order), B, H, dh, max_len, dh ** -0.5, window, products, ptr(out), _stream(qkv)),
              "mr_attn_split_f32")
    else:
        check(lib.mr_attn_f32(ptr(qkv), ptr(cu_seqlens), ptr(seq_order), B, H, dh, max_len, dh ** -0.5, window, ptr(out), _stream(qkv)), "mr_attn_f32")
    PROF.end(ev, qkv.device, {0: "attention", 3: "attention_bf16x3", 6: "attention_bf16x6", PRODUCTS_F16X3: "attention_f16x3"}[products],
             flops=ATTN_FLOPS_HINT[0], nbytes=4.0 * T * 4 * H * dh)
    ATTN_FLOPS_HINT[0] = 0.0
    return out


def attention_global_row(qg: torch.Tensor, kvg: torch.Tensor, cu_seqlens: torch.Tensor, B: int, H: int, max_len: int, ctx: torch.Tensor,
                         compact: bool = False, drop_p: float = 0.0, drop_key: int = 0):
    """Longformer global row into ctx[cu[b]] (ctx (T, d)) or, compact, into row b of a (B, d) matrix.  drop_p > 0: training-graph dropout
    on the row's probabilities."""
    dh = qg.shape[1] // H
    if drop_p > 0.0:
        check(_lib.load().mr_attn_global_row_train_f32(ptr(qg), ptr(kvg), ptr(cu_seqlens), B, H, dh, max_len, dh ** -0.5, drop_p, drop_key, ptr(ctx),
                                                       int(compact), _stream(qg)), "mr_attn_global_row_train_f32")
        return ctx
    check(_lib.load().mr_attn_global_row_f32(ptr(qg), ptr(kvg), ptr(cu_seqlens), B, H, dh, max_len, dh ** -0.5, ptr(ctx), int(compact), _stream(qg)),
          "mr_attn_global_row_f32")
    return ctx


def cls_pool_normalize(x: torch.Tensor, cu_seqlens: torch.Tensor, B: int, normalize: bool, out=None) -> torch.Tensor:
    d = x.shape[1]
    out = torch.empty(B, d, dtype=torch.float32, device=x.device) if out is None else out
    check(_lib.load().mr_cls_pool_normalize_f32(ptr(x), x.stride(0), ptr(cu_seqlens), B, d, int(normalize), ptr(out), _stream(x)), "mr_cls_pool_normalize_f32")
    return out


def mean_pool(x: torch.Tensor, cu_seqlens: torch.Tensor, xpad: torch.Tensor, pad_len: torch.Tensor, B: int, normalize: bool) -> torch.Tensor:
    """(B, d) mean over each sequence's PADDED width: packed token rows of ``x`` plus (pad_len[b] - len_b) copies of ``xpad[b]`` (the pad
    positions' shared hidden state), divided by pad_len[b]; L2-normalised if asked."""
    _dev(xpad, "xpad", torch.float32), _dev(pad_len, "pad_len", torch.int32), _dev(cu_seqlens, "cu_seqlens", torch.int32)
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1) or xpad.shape != (B, x.shape[1]) or pad_len.numel() != B:
        raise ValueError("x (T, d) fp32 GPU rows, xpad (B, d), pad_len (B,)")
    out = torch.empty(B, x.shape[1], dtype=torch.float32, device=x.device)
    check(_lib.load().mr_mean_pool_f32(ptr(x), x.stride(0), ptr(cu_seqlens), ptr(xpad), ptr(pad_len), B, x.shape[1], int(normalize), ptr(out),
                                       _stream(x)), "mr_mean_pool_f32")
    return out


def gather_rows(x: torch.Tensor, row_idx: torch.Tensor, out=None) -> torch.Tensor:
    n, d = row_idx.numel(), x.shape[1]
    out = torch.empty(n, d, dtype=torch.float32, device=x.device) if out is None else out
    check(_lib.load().mr_gather_rows_f32(ptr(x), x.stride(0), ptr(row_idx), n, d, ptr(out), out.stride(0), _stream(x)), "mr_gather_rows_f32")
    return out


# ------------------------------------------------------------------------------------------ scoring
def topk_rows(scores: torch.Tensor, k: int, labels: Optional[torch.Tensor] = None, inv_temp: float = 1.0):
    """``scores``: (R, C) fp32 with unit column stride (rows may be padded: a leading dimension that is a multiple of 4 on a 16-byte
    aligned block lets the kernel keep each row in LDS)."""
    if not (isinstance(scores, torch.Tensor) and scores.is_cuda and scores.dtype == torch.float32 and scores.dim() == 2 and scores.stride(1) == 1):
        raise ValueError("scores must be a (R, C) fp32 GPU matrix with unit column stride (the HIP path has no CPU fallback)")
    R, C = scores.shape
    dev = scores.device
    val = torch.empty(R, k, dtype=torch.float32, device=dev)
    idx = torch.empty(R, k, dtype=torch.int64, device=dev)
    lse = lab = rank = None
    if labels is not None:
        _dev(labels, "labels", torch.int64)
        lse = torch.empty(R, dtype=torch.float32, device=dev)
        lab = torch.empty(R, dtype=torch.float32, device=dev)
        rank = torch.empty(R, dtype=torch.int32, device=dev)
    ev = PROF.begin(dev)
    check(_lib.load().mr_topk_rows_f32(ptr(scores), scores.stride(0), R, C, k, ptr(val), ptr(idx), ptr(labels), inv_temp, ptr(lse), ptr(lab), ptr(rank), _stream(scores)), "mr_topk_rows_f32")
    PROF.end(ev, dev, "topk_rows", flops=0.0, nbytes=4.0 * R * C)  # algorithmic: every score read once
    return val, idx, lse, lab, rank


def score_topk(U: torch.Tensor, E: torch.Tensor, k: int, labels: Optional[torch.Tensor] = None, inv_temp: float = 1.0,
               return_scores: bool = False, fused: Optional[bool] = None):
    """scores = U @ E.T -> canonical top-k (+ lse / label logit / label rank when labels are given).  Without ``return_scores`` the
    library picks the route (include/mergerec_hip.h, mr_score_fused_mode): selection inside the scoring kernel (csrc/score_fused.hip: the
    (users x M) block is never written, only per-part candidates reach memory) when the block would not stay cache-resident, the scoring
    GEMM into a workspace + row select otherwise.  ``fused`` True / False forces a route for this call."""
    _dev(U, "U", torch.float32), _dev(E, "E", torch.float32)
    nU, d = U.shape
    M = E.shape[0]
    dev = U.device
    lib = _lib.load()
    val = torch.empty(nU, k, dtype=torch.float32, device=dev)
    idx = torch.empty(nU, k, dtype=torch.int64, device=dev)
    lse = lab = rank = None
    if labels is not None:
        lse = torch.empty(nU, dtype=torch.float32, device=dev)
        lab = torch.empty(nU, dtype=torch.float32, device=dev)
        rank = torch.empty(nU, dtype=torch.int32, device=dev)
    staged = return_scores or fused is False or (fused is None and lib.mr_score_topk_ws_bytes_ex(nU, M, d, k) >= 4 * nU * M)
    if PROF.enabled and staged:
        # profiling the staged route: its two launches timed separately (scoring GEMM on the fp32 matrix cores, then the row select) --
        # identical results
        ldm = (M + 3) // 4 * 4  # the entry point's workspace layout
        sc = torch.empty(nU, ldm, dtype=torch.float32, device=dev)[:, :M]
        gemm_nt(U, [E], out=sc, prof_name="score_gemm")
        PROF.chain()  # the row select follows the scoring GEMM directly (its output tensors are allocations, not launches)
        val, idx, lse, lab, rank = topk_rows(sc, k, labels, inv_temp)
        return val, idx, lse, lab, rank, (sc.contiguous() if return_scores else None)
    scores = ws = None
    nbytes = 0
    prev = lib.mr_score_fused_mode(-1 if fused is None else int(bool(fused)))
    try:
        if return_scores:
            scores = torch.empty(nU, M, dtype=torch.float32, device=dev)
        else:
            nbytes = lib.mr_score_topk_ws_bytes_ex(nU, M, d, k)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ev = PROF.begin(dev)
        check(lib.mr_score_topk_f32(ptr(U), ptr(E), nU, M, d, k, ptr(val), ptr(idx), ptr(scores), ptr(labels), inv_temp, ptr(lse), ptr(lab), ptr(rank), ptr(ws), nbytes, _stream(U)), "mr_score_topk_f32")
        PROF.end(ev, dev, "score_topk", flops=2.0 * nU * M * d, nbytes=4.0 * (nU + M) * d + nU * k * 12)
    finally:
        if fused is not None:
            lib.mr_score_fused_mode(prev)
    return val, idx, lse, lab, rank, scores


# ------------------------------------------------------------------------------------------ distillation losses (next-row 2)
def skinny_scores(reps: torch.Tensor, E: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[i][m] = <reps[i], E[m]> for a handful of rows (any count: 8 per launch) against a whole catalog: one stream over E."""
    _dev(reps, "reps", torch.float32), _dev(E, "E", torch.float32)
    n, d = reps.shape
    M = E.shape[0]
    if E.shape[1] != d or reps.stride(1) != 1 or E.stride(1) != 1:
        raise ValueError("reps (n, d) and E (M, d) with unit inner stride expected")
    out = torch.empty(n, M, dtype=torch.float32, device=reps.device) if out is None else out
    lib = _lib.load()
    for i0 in range(0, n, 8):
        k = min(8, n - i0)
        check(lib.mr_skinny_scores_f32(ptr(reps[i0:]), reps.stride(0), k, ptr(E), E.stride(0), M, d, ptr(out[i0:]), out.stride(0), _stream(reps)),
              "mr_skinny_scores_f32")
    return out


def skinny_scores_bwd(dz: torch.Tensor, E: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """d_reps[i][:] = scale * sum_m dz[i][m] E[m][:] (dz (n, M) with unit inner stride, possibly a view of wider rows)."""
    _dev(E, "E", torch.float32)
    if not (isinstance(dz, torch.Tensor) and dz.is_cuda and dz.dtype == torch.float32 and dz.dim() == 2):
        raise ValueError("dz must be a 2-D fp32 GPU tensor (the HIP path has no CPU fallback)")
    n, M = dz.shape
    d = E.shape[1]
    if E.shape[0] != M or dz.stride(1) != 1:
        raise ValueError("dz (n, M) and E (M, d) expected")
    out = torch.empty(n, d, dtype=torch.float32, device=dz.device) if out is None else out
    lib = _lib.load()
    for i0 in range(0, n, 8):
        k = min(8, n - i0)
        nbytes = lib.mr_skinny_bwd_ws_bytes(k, M, d)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dz.device)
        check(lib.mr_skinny_bwd_f32(ptr(dz[i0:]), dz.stride(0), k, ptr(E), E.stride(0), M, d, float(scale), ptr(out[i0:]), ptr(ws), nbytes, _stream(dz)),
              "mr_skinny_bwd_f32")
    return out


def distill_loss_rows(z: torch.Tensor, t: Optional[torch.Tensor], *, label_src: int = 0, w_ce: float = 0.0, w_kd: float = 0.0,
                      temperature: float = 1.0, w_ent: float = 0.0, w_mse: float = 0.0, w_pair: float = 0.0, margin: float = 0.0,
                      w_listnet: float = 0.0, want_grad: bool = False, grad_scale: float = 1.0, dz: Optional[torch.Tensor] = None,
                      row_M: Optional[torch.Tensor] = None):
    """Per-row loss (rows,) and, if asked, grad_scale * d loss_row / d z (rows, M); see mr_distill_loss_rows_f32.
    ``dz`` may be a caller-owned (rows, M) view with a padded leading dimension.  ``row_M`` (device int32, rows): row r holds only
    row_M[r] <= M logits -- the rows of several catalogs in one launch (mr_distill_loss_rows_var_f32)."""
    _dev(z, "z", torch.float32)
    if z.dim() != 2 or z.stride(1) != 1:
        raise ValueError("z must be a (rows, M) matrix with unit column stride")
    rows, M = z.shape
    if t is not None:
        _dev(t, "t", torch.float32)
        if t.shape != z.shape or t.stride(1) != 1:
            raise ValueError("t must match z")
    loss_row = torch.empty(rows, dtype=torch.float32, device=z.device)
    if dz is None and want_grad:
        dz = torch.empty(rows, M, dtype=torch.float32, device=z.device)
    if dz is not None and (dz.shape != z.shape or dz.stride(1) != 1):
        raise ValueError("dz must match z")
    ev = PROF.begin(z.device)
    if row_M is not None:
        check(_lib.load().mr_distill_loss_rows_var_f32(ptr(z), z.stride(0), ptr(t), t.stride(0) if t is not None else 0, rows, M, ptr(row_M), label_src,
                                                       w_ce, w_kd, temperature, w_ent, w_mse, w_pair, margin, w_listnet, ptr(loss_row), ptr(dz),
                                                       dz.stride(0) if dz is not None else 0, grad_scale, _stream(z)), "mr_distill_loss_rows_var_f32")
    else:
        check(_lib.load().mr_distill_loss_rows_f32(ptr(z), z.stride(0), ptr(t), t.stride(0) if t is not None else 0, rows, M, label_src, w_ce,
                                                   w_kd, temperature, w_ent, w_mse, w_pair, margin, w_listnet, ptr(loss_row), ptr(dz),
                                                   dz.stride(0) if dz is not None else 0, grad_scale, _stream(z)), "mr_distill_loss_rows_f32")
    PROF.end(ev, z.device, "distill_loss_rows", flops=0.0, nbytes=4.0 * rows * M * (2 + (1 if dz is not None else 0)))
    return loss_row, dz


# ------------------------------------------------------------------------------------------ encoder backward (merge_train)
def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


@functools.lru_cache(maxsize=4096)
def splitk_plan(M: int, N: int, K: int, has_residual: bool = False) -> int:
    """Number of k slices for the exact-fp32 training product (M, K) x (N, K)^T: the argmin of a launch model fitted to
    ``tools/splitk_sweep.py`` on an MI355X (it ranks every measured shape's candidates in the measured order).  These products are
    latency-bound -- one workgroup per CU runs a 128 x 128 x 16 tile step in ~1.2 us, 2.3 x the matrix pipe's time -- so a launch costs
    (rounds of 256 workgroups) x (tile steps per workgroup), and cutting k pays until the partial sums' traffic (s x M x N x 4 B written
    and read again) and the second launch outweigh it.  One slice goes through the plain kernel, which halves the tile height when that
    fills the chip better (no residual).  A pure function of the shape: the same slices, hence the same bits, on every rank and run."""
    nk = max(K // 16, 1)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    step_us, launch_us, bytes_per_us = 1.2, 6.0, 8.0e6  # the partial sums are re-read out of L2 / Infinity Cache
    half_tiles = ((M + 63) // 64) * ((N + 127) // 128)
    if not has_residual and -(-half_tiles // 256) * 0.5 * 1.05 < -(-tiles // 256):  # csrc/gemm.hip's own rule
        best, best_cost = 1, -(-half_tiles // 256) * nk * 0.5 * step_us
    else:
        best, best_cost = 1, -(-tiles // 256) * nk * step_us
    for sp in range(2, min(nk, 16) + 1):
        chunk = -(-nk // sp)
        if -(-nk // chunk) != sp:  # the library rounds slices to whole k tiles: this count collapses to a smaller one
            continue
        cost = -(-tiles * sp // 256) * chunk * step_us + launch_us + sp * M * N * 8.0 / bytes_per_us
        if cost < best_cost:
            best, best_cost = sp, cost
    return best


def gemm_nt_train(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, splits: Optional[int] = None) -> torch.Tensor:
    """A @ W.T (+ bias) (+ residual) for the training graph: split-K when the output has too few tiles to fill the chip."""
    if not (A.is_cuda and W.is_cuda and A.dtype == torch.float32 and W.dtype == torch.float32):
        raise ValueError("A and W must be fp32 GPU tensors (the HIP path has no CPU fallback)")
    M, K = A.shape
    N = W.shape[0]
    if W.shape[1] != K or not W.is_contiguous() or A.stride(1) != 1:
        raise ValueError("A (M, K) with unit inner stride and contiguous W (N, K) expected")
    out = torch.empty(M, N, dtype=torch.float32, device=A.device) if out is None else out
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if splits is None:
        splits = splitk_plan(M, N, K, residual is not None)
    lib = _lib.load()
    nbytes = lib.mr_gemm_nt_splitk_ws_bytes(M, N, splits)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=A.device) if splits > 1 else None
    ev = PROF.begin(A.device)
    check(lib.mr_gemm_nt_splitk_f32(ptr(A), A.stride(0), ptr(W), ptr(bias), M, N, K, ptr(residual), 0 if residual is None else residual.stride(0),
                                    ptr(out), out.stride(0), splits, ptr(ws), nbytes, _stream(A)), "mr_gemm_nt_splitk_f32")
    PROF.end(ev, A.device, "gemm_nt_train", flops=2.0 * M * N * K, nbytes=4.0 * (M * K + N * K + M * N))
    return out


EPI_NONE, EPI_GELU_FWD, EPI_GELU_BWD = 0, 1, 2


def gemm_tile(A: torch.Tensor, Bs, trans_a: bool = False, trans_b: bool = False, biases=None, residual: Optional[torch.Tensor] = None, out=None,
              colsum=None, epi: int = EPI_NONE, E: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None, drop_p: float = 0.0,
              drop_key: int = 0, bn: int = 0, products: int = 0) -> torch.Tensor:
    """Token-sized exact-fp32 product of the training graph (mr_gemm_tile_f32): ``Aop @ Bop^T`` with either operand in either orientation,
    no transposed copies, no split-K; every output element is the ascending-k FMA chain.
      A: (M, K), or with trans_a (K, M).   Bs: 1..3 matrices -- (seg_n, K) each, stacked along N; with trans_b (seg_k, N) each, stacked along K.
      out: an (M, N) tensor, or (trans_a only) a list of 1..3 (seg_m, N) tensors stacked along M with optional ``colsum`` (seg_m,) each.
      epilogue: + bias (per N segment; one bias with trans_b), dropout (drop_p, drop_key; row = m, column = n), + residual,
      EPI_GELU_FWD (out = pre-activation, out2 = gelu) or EPI_GELU_BWD (out = product * gelu'(E)).
      products: 0 = exact fp32 (the FMA chain), 6 = bf16x6 split precision (fp32-grade, 2.7 x fewer matrix-pipe cycles)."""
    Bs = list(Bs)
    if not (A.is_cuda and A.dtype == torch.float32 and A.dim() == 2 and A.stride(1) == 1):
        raise ValueError("A must be a 2-D fp32 GPU tensor with unit inner stride (the HIP path has no CPU fallback)")
    (K, M) = A.shape if trans_a else A.shape[::-1]
    nb = len(Bs)
    if trans_b:
        N, seg_b = Bs[0].shape[1], Bs[0].shape[0]
        if sum(b.shape[0] for b in Bs) != K:
            raise ValueError("B segments must cover K")
    else:
        seg_b, N = Bs[0].shape[0], Bs[0].shape[0] * nb
        if any(b.shape[1] != K for b in Bs):
            raise ValueError("every B segment must be (seg_n, K)")
    if any(b.stride(1) != 1 or b.stride(0) != Bs[0].stride(0) or b.shape != Bs[0].shape for b in Bs):
        raise ValueError("B segments must share one shape and row pitch")
    outs = list(out) if isinstance(out, (list, tuple)) else [out if out is not None else torch.empty(M, N, dtype=torch.float32, device=A.device)]
    nc = len(outs)
    seg_c = outs[0].shape[0]
    if any(o.shape != (seg_c, N) or o.stride(1) != 1 or o.stride(0) != outs[0].stride(0) for o in outs) or seg_c * nc != M:
        raise ValueError("out must be (M, N), or equally shaped segments covering M")
    biases = list(biases) if biases is not None else []
    biases += [None] * (3 - len(biases))
    cs = list(colsum) if colsum is not None else []
    cs += [None] * (3 - len(cs))
    Bp = Bs + [None] * (3 - nb)
    Cp = outs + [None] * (3 - nc)
    ev = PROF.begin(A.device)
    check(_lib.load().mr_gemm_tile_f32(
        ptr(A), A.stride(0) if A.shape[0] > 1 else A.shape[1], int(trans_a), ptr(Bp[0]), ptr(Bp[1]), ptr(Bp[2]),
        Bs[0].stride(0) if Bs[0].shape[0] > 1 else Bs[0].shape[1], int(trans_b), nb, seg_b, ptr(biases[0]), ptr(biases[1]),
        ptr(biases[2]), M, N, K, ptr(residual), 0 if residual is None else residual.stride(0), ptr(Cp[0]), ptr(Cp[1]), ptr(Cp[2]), outs[0].stride(0), nc,
        seg_c, ptr(cs[0]), ptr(cs[1]), ptr(cs[2]), epi, ptr(E), 0 if E is None else E.stride(0), ptr(out2), 0 if out2 is None else out2.stride(0),
        float(drop_p), int(drop_key), products, bn, _stream(A)), "mr_gemm_tile_f32")
    PROF.end(ev, A.device, "gemm_tile", flops=2.0 * M * N * K, nbytes=4.0 * (M * K + N * K + M * N))
    return outs[0] if nc == 1 else outs


def transpose_pad(x: torch.Tensor, out: Optional[torch.Tensor] = None, pad: int = 16) -> torch.Tensor:
    """(R, C) -> (C, R padded to a multiple of ``pad``) with the pad columns zeroed: the K-contiguous operand layout of the NT GEMM."""
    _dev(x, "x", torch.float32)
    R, C = x.shape
    Rp = (R + pad - 1) // pad * pad
    out = torch.empty(C, Rp, dtype=torch.float32, device=x.device) if out is None else out
    check(_lib.load().mr_transpose_f32(ptr(x), x.stride(0), R, C, ptr(out), out.stride(0), Rp, _stream(x)), "mr_transpose_f32")
    return out


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _dev(x, "x", torch.float32)
    R, C = x.shape
    out = torch.empty(C, dtype=torch.float32, device=x.device) if out is None else out
    lib = _lib.load()
    nbytes = lib.mr_colsum_ws_bytes(R, C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    check(lib.mr_colsum_f32(ptr(x), x.stride(0), R, C, ptr(out), ptr(ws), nbytes, _stream(x)), "mr_colsum_f32")
    return out


def rowsum(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[r] = sum_c x[r][c] for a (R, C) matrix with unit inner stride."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise ValueError("x must be a (R, C) fp32 GPU matrix with unit inner stride (the HIP path has no CPU fallback)")
    R, C = x.shape
    out = torch.empty(R, dtype=torch.float32, device=x.device) if out is None else out
    check(_lib.load().mr_rowsum_f32(ptr(x), x.stride(0), R, C, ptr(out), _stream(x)), "mr_rowsum_f32")
    return out


def gelu_fwd(u: torch.Tensor) -> torch.Tensor:
    _dev(u, "u", torch.float32)
    if not u.is_contiguous():
        raise ValueError("u must be contiguous")
    h = torch.empty_like(u)
    check(_lib.load().mr_gelu_fwd_f32(ptr(u), u.numel(), ptr(h), _stream(u)), "mr_gelu_fwd_f32")
    return h


def gelu_bwd(u: torch.Tensor, dh: torch.Tensor) -> torch.Tensor:
    _dev(u, "u", torch.float32), _dev(dh, "dh", torch.float32)
    if not (u.is_contiguous() and dh.is_contiguous() and u.shape == dh.shape):
        raise ValueError("u and dh must be contiguous and equally shaped")
    du = torch.empty_like(u)
    check(_lib.load().mr_gelu_bwd_f32(ptr(u), ptr(dh), u.numel(), ptr(du), _stream(u)), "mr_gelu_bwd_f32")
    return du


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, eps: float, dgamma: Optional[torch.Tensor] = None,
                  dbeta: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx of LayerNorm(x) given dy; writes the parameter gradients into dgamma / dbeta when given."""
    _dev(x, "x", torch.float32), _dev(dy, "dy", torch.float32)
    T, d = x.shape
    dx = torch.empty(T, d, dtype=torch.float32, device=x.device)
    stats = torch.empty(max(T, 1), 2, dtype=torch.float32, device=x.device)
    lib = _lib.load()
    nbytes = lib.mr_layernorm_bwd_ws_bytes(T, d) if dgamma is not None else 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if nbytes else None
    check(lib.mr_layernorm_bwd_f32(ptr(x), x.stride(0), ptr(dy), dy.stride(0), ptr(gamma), eps, T, d, ptr(dx), dx.stride(0), ptr(stats),
                                   ptr(dgamma), ptr(dbeta), ptr(ws), nbytes, _stream(x)), "mr_layernorm_bwd_f32")
    return dx


def attention_bwd(qkv: torch.Tensor, ctx: torch.Tensor, dctx: torch.Tensor, cu_seqlens: torch.Tensor, B: int, H: int, scale: Optional[float] = None,
                  window: int = -1, max_len: Optional[int] = None, seq_order: Optional[torch.Tensor] = None, drop_p: float = 0.0, drop_key: int = 0,
                  work=None):
    _dev(qkv, "qkv", torch.float32), _dev(ctx, "ctx", torch.float32), _dev(dctx, "dctx", torch.float32)
    T = qkv.shape[0]
    if not (qkv.is_contiguous() and ctx.is_contiguous() and dctx.is_contiguous()) or qkv.shape[1] != 3 * H * 64:
        raise ValueError("qkv (T, 3 H 64), ctx / dctx (T, H 64) must be contiguous")
    dqkv = torch.empty_like(qkv)
    rowstat = torch.empty(max(T, 1), H, 2, dtype=torch.float32, device=qkv.device)
    if max_len is None:  # (one small D2H sync; the training graph passes the packed batch's own maximum)
        max_len = int((cu_seqlens[1:] - cu_seqlens[:-1]).max()) if B else 0
    ev = PROF.begin(qkv.device)
    wl = work.get(128) if work else None
    if wl is not None and os.environ.get("MR_ATTN_WORKLIST", "1") != "0":  # the (sequence, 128-row block) pairs that exist: no empty workgroups
        check(_lib.load().mr_attn_bwd_work_f32(ptr(qkv), ptr(ctx), ptr(dctx), ptr(cu_seqlens), ptr(wl[0]), wl[1], B, 128, H, 64, 0.125 if scale is None else scale,
                                               window, float(drop_p), drop_key, ptr(rowstat), ptr(dqkv), _stream(qkv)), "mr_attn_bwd_work_f32")
    elif drop_p > 0.0:
        check(_lib.load().mr_attn_bwd_train_f32(ptr(qkv), ptr(ctx), ptr(dctx), ptr(cu_seqlens), ptr(seq_order), B, H, 64, max_len,
                                                0.125 if scale is None else scale, window, drop_p, drop_key, ptr(rowstat), ptr(dqkv), _stream(qkv)),
              "mr_attn_bwd_train_f32")
    else:
        check(_lib.load().mr_attn_bwd_f32(ptr(qkv), ptr(ctx), ptr(dctx), ptr(cu_seqlens), ptr(seq_order), B, H, 64, max_len, 0.125 if scale is None else scale, window,
                                          ptr(rowstat), ptr(dqkv), _stream(qkv)), "mr_attn_bwd_f32")
    PROF.end(ev, qkv.device, "attention_bwd", flops=0.0, nbytes=4.0 * T * H * 64 * 8)
    return dqkv


def attention_global_row_bwd(qg: torch.Tensor, kvg: torch.Tensor, ctx_cls: torch.Tensor, dctx_cls: torch.Tensor, cu_seqlens: torch.Tensor, B: int, H: int,
                             drop_p: float = 0.0, drop_key: int = 0):
    """-> (dqg (B, H 64), dkvg (T, 2 H 64)) of the Longformer global row."""
    for t, n in ((qg, "qg"), (kvg, "kvg"), (ctx_cls, "ctx_cls"), (dctx_cls, "dctx_cls")):
        _dev(t, n, torch.float32)
        if not t.is_contiguous():
            raise ValueError(f"{n} must be contiguous")
    dqg, dkvg = torch.empty_like(qg), torch.empty_like(kvg)
    if drop_p > 0.0:
        check(_lib.load().mr_attn_global_row_bwd_train_f32(ptr(qg), ptr(kvg), ptr(ctx_cls), ptr(dctx_cls), ptr(cu_seqlens), B, H, 64, 0.125, drop_p,
                                                           drop_key, ptr(dqg), ptr(dkvg), _stream(qg)), "mr_attn_global_row_bwd_train_f32")
        return dqg, dkvg
    check(_lib.load().mr_attn_global_row_bwd_f32(ptr(qg), ptr(kvg), ptr(ctx_cls), ptr(dctx_cls), ptr(cu_seqlens), B, H, 64, 0.125, ptr(dqg), ptr(dkvg),
                                                 _stream(qg)), "mr_attn_global_row_bwd_f32")
    return dqg, dkvg


# ------------------------------------------------------------------------------------------ training-graph dropout (csrc/dropout.h)
DROP_SITE_EMBED, DROP_SITE_ATTN_PROBS, DROP_SITE_ATTN_OUT, DROP_SITE_FFN_OUT, DROP_SITE_GLOBAL_ROW = range(5)


def _lowbias32(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def dropout_site_key(seed: int, step: int, layer: int, site: int) -> int:
    """== mr_dropout_site_key (host arithmetic on both sides; tests/test_cabi_symbols.py holds them together)."""
    return _lowbias32(_lowbias32(_lowbias32(seed) + step) + layer * 8 + site)


def dropout_rows(x: torch.Tensor, p: float, key: int, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = dropout(x) (+ residual) with the mask of csrc/dropout.h (row = x's row index, col = column); the same call on dY is the
    site's backward.  p == 0: x (+ residual) unchanged."""
    _dev(x, "x", torch.float32)
    T, d = x.shape
    out = torch.empty(T, d, dtype=torch.float32, device=x.device) if out is None else out
    check(_lib.load().mr_dropout_rows_f32(ptr(x), x.stride(0), T, d, float(p), key, ptr(residual), residual.stride(0) if residual is not None else 0,
                                          ptr(out), out.stride(0), _stream(x)), "mr_dropout_rows_f32")
    return out


def scatter_add_rows(src: torch.Tensor, idx: torch.Tensor, table: torch.Tensor):
    _dev(src, "src", torch.float32), _dev(idx, "idx", torch.int32), _dev(table, "table", torch.float32)
    T, d = src.shape
    check(_lib.load().mr_scatter_add_rows_f32(ptr(src), src.stride(0), ptr(idx), T, d, ptr(table), table.stride(0), _stream(src)),
          "mr_scatter_add_rows_f32")
    return table


# ------------------------------------------------------------------------------------------ optimizer step (finetune_train)
def sum_squares(x: torch.Tensor) -> torch.Tensor:
    """||x||^2 of a flat arena vector as a 1-element DEVICE tensor: the deterministic two-stage dot-product reduction of the
    alpha-gradient kernel with the vector as its own 'task vector' (feeds the clip coefficient of ``adamw_step`` without a host sync)."""
    return merge_bwd_alpha(x.view(1, -1), x).view(1)


def adamw_step(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, *, lr: float, step: int,
               betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, seg_off: Optional[torch.Tensor] = None,
               seg_wd: Optional[torch.Tensor] = None, grad_sumsq: Optional[torch.Tensor] = None, max_grad_norm: float = 0.0):
    """One fused AdamW step over flat fp32 arenas, in place (see mr_adamw_step_f32)."""
    for t, n in ((param, "param"), (grad, "grad"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        _dev(t, n, torch.float32)
        if t.dim() != 1 or t.numel() != param.numel():
            raise ValueError(f"{n} must be a flat vector of the arena's length")
    S = 0
    if seg_off is not None:
        _dev(seg_off, "seg_off", torch.int64), _dev(seg_wd, "seg_wd", torch.float32)
        S = seg_wd.numel()
        if seg_off.numel() != S + 1:
            raise ValueError("seg_off needs one more entry than seg_wd")
    if grad_sumsq is not None:
        _dev(grad_sumsq, "grad_sumsq", torch.float32)
    n = param.numel()
    ev = PROF.begin(param.device)
    check(_lib.load().mr_adamw_step_f32(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), n, ptr(seg_off), ptr(seg_wd), S, float(lr),
                                        float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), ptr(grad_sumsq),
                                        float(max_grad_norm), _stream(param)), "mr_adamw_step_f32")
    PROF.end(ev, param.device, "adamw_step", flops=0.0, nbytes=28.0 * n)
    return param


# ------------------------------------------------------------------------------------------ bf16x3 products for the training graph
_KB_TABLES = {}


def split_matrix_kblock(x: torch.Tensor, pieces=None):
    """One contiguous fp32 (N, K) matrix -> its (hi, mid, lo) bf16 pieces in the k-blocked layout the split GEMMs read
    (``[K / 16][N][16]``).  K % 16 == 0."""
    _dev(x, "x", torch.float32)
    if x.dim() != 2 or not x.is_contiguous():
        raise ValueError("x must be a contiguous (N, K) matrix")
    N, K = x.shape
    key = (N, K, x.device)
    table = _KB_TABLES.get(key)
    if table is None:
        table = _KB_TABLES[key] = KBlockTable([(0, N, K)], x.device)
    if pieces is None:
        pieces = tuple(torch.empty(N * K, dtype=torch.bfloat16, device=x.device) for _ in range(3))
    return split_weights_kblock(x.view(-1), table, pieces)


def split_tokens_kblock(x: torch.Tensor, pad: int = 32):
    """Token-major x (T, C) -> ((hi, mid) bf16 pieces of x^T in k-blocked form, T_pad): the weight-gradient operand without the fp32
    transpose (mr_split_tokens_kblock_f32)."""
    _dev(x, "x", torch.float32)
    if x.dim() != 2 or x.stride(1) != 1:
        raise ValueError("x must be (T, C) with unit inner stride")
    T, C = x.shape
    Tp = (T + pad - 1) // pad * pad
    hi, mid = (torch.empty(C * Tp, dtype=torch.bfloat16, device=x.device) for _ in range(2))
    ev = PROF.begin(x.device)
    check(_lib.load().mr_split_tokens_kblock_f32(ptr(x), x.stride(0), T, C, Tp, ptr(hi), ptr(mid), _stream(x)), "mr_split_tokens_kblock_f32")
    PROF.end(ev, x.device, "split_tokens", nbytes=8.0 * T * C)
    return (hi, mid), Tp


def gemm_nt_split_k(A: torch.Tensor, pieces, off: int, N: int, K: int, bias: Optional[torch.Tensor] = None,
                    residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, splits: Optional[int] = None) -> torch.Tensor:
    """A (M, K) @ W.T for ONE pre-split k-blocked weight (N, K) at element offset ``off`` of ``pieces``, bf16x3, with K split over
    workgroups when the output alone cannot fill the chip (weight gradients: K = tokens)."""
    if A.dim() != 2 or A.stride(1) != 1 or A.shape[1] != K:
        raise ValueError("A must be (M, K) with unit inner stride")
    M = A.shape[0]
    out = torch.empty(M, N, dtype=torch.float32, device=A.device) if out is None else out
    if splits is None:
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        splits = max(1, min(K // 64, 64, 1024 // max(tiles, 1)))
    lib = _lib.load()
    nbytes = lib.mr_gemm_nt_bf16x3_splitk_ws_bytes(M, N, splits)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=A.device) if splits > 1 else None
    ev = PROF.begin(A.device)
    check(lib.mr_gemm_nt_bf16x3_splitk_f32(ptr(A), A.stride(0), ptr(pieces[0]), ptr(pieces[1]), off, ptr(bias), M, N, K, ptr(residual),
                                           0 if residual is None else residual.stride(0), ptr(out), out.stride(0), splits, ptr(ws), nbytes,
                                           _stream(A)), "mr_gemm_nt_bf16x3_splitk_f32")
    PROF.end(ev, A.device, "gemm_nt_bf16x3_train", flops=2.0 * M * N * K, nbytes=4.0 * (M * K + M * N) + 4.0 * N * K)
    return out
