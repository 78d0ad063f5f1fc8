"""ctypes loader for libmergerec_hip.so -- the only door between the Python host layer and the HIP kernels.

The product path has NO CPU or eager-torch fallback: if the library is missing or a call returns a
non-zero code, a MergeRecHipError is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from pathlib import Path

import torch  # noqa: F401  (must be imported first: the .so then binds to torch's HIP runtime, SONAME libamdhip64.so.7)

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libmergerec_hip.so"
HEADER_PATH = _HERE.parent / "include" / "mergerec_hip.h"


class MergeRecHipError(RuntimeError):
    pass


c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float
c_d = ctypes.c_double
c_sz = ctypes.c_size_t
c_u32 = ctypes.c_uint32

# name -> (restype, argtypes); mirrors include/mergerec_hip.h
SIGNATURES = {
    "mr_version": (c_i, []),
    "mr_strerror": (ctypes.c_char_p, [c_i]),
    "mr_last_hip_error": (ctypes.c_char_p, []),
    "mr_task_vector_f32": (c_i, [c_p, c_p, c_i64, c_p, c_p]),
    "mr_merge_nway_f32": (c_i, [c_p, c_p, c_i64, c_p, c_p, c_i, c_i, c_i64, c_i64, c_p, c_p]),
    "mr_merge_running_f32": (c_i, [c_p, c_p, c_i64, c_p, c_i, c_i64, c_p, c_p]),
    "mr_merge_rows_f32": (c_i, [c_p, c_p, c_i64, c_p, c_i, c_p, c_i, c_i, c_i, c_i64, c_p, c_p]),
    "mr_merge_bwd_alpha_ws_bytes": (c_sz, [c_i, c_i, c_i64]),
    "mr_merge_bwd_alpha_f32": (c_i, [c_p, c_i64, c_p, c_p, c_i, c_i, c_i64, c_p, c_p, c_sz, c_p]),
    "mr_select_ws_bytes": (c_sz, [c_i64]),
    "mr_abs_kth_largest_f32": (c_i, [c_p, c_i64, c_i64, c_p, c_p, c_p, c_sz, c_p]),
    "mr_abs_topk_mask_f32": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "mr_ties_combine_f32": (c_i, [c_p, c_i64, c_i, c_i64, c_p]),
    "mr_lns_combine_f32": (c_i, [c_p, c_p, c_i64, c_i, c_i64, c_p, c_p]),
    "mr_kth_largest_value_f32": (c_i, [c_p, c_i64, c_i64, c_i, c_p, c_p, c_sz, c_p]),
    "mr_pcb_stage1_f32": (c_i, [c_p, c_i64, c_i, c_i, c_i64, c_p, c_p, c_p, c_p]),
    "mr_pcb_stage2_f32": (c_i, [c_p, c_p, c_i64, c_i, c_i64, c_p, c_p, c_p]),
    "mr_distill_loss_rows_f32": (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_i64, c_f, c_p]),
    "mr_gemm_nt_splitk_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "mr_gemm_nt_splitk_f32": (c_i, [c_p, c_i64, c_p, c_p, c_i, c_i, c_i, c_p, c_i64, c_p, c_i64, c_i, c_p, c_sz, c_p]),
    "mr_transpose_f32": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_i64, c_i, c_p]),
    "mr_rowsum_f32": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p]),
    "mr_colsum_ws_bytes": (c_sz, [c_i, c_i]),
    "mr_colsum_f32": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p, c_sz, c_p]),
    "mr_gelu_fwd_f32": (c_i, [c_p, c_i64, c_p, c_p]),
    "mr_gelu_bwd_f32": (c_i, [c_p, c_p, c_i64, c_p, c_p]),
    "mr_layernorm_bwd_ws_bytes": (c_sz, [c_i, c_i]),
    "mr_layernorm_bwd_f32": (c_i, [c_p, c_i64, c_p, c_i64, c_p, c_f, c_i, c_i, c_p, c_i64, c_p, c_p, c_p, c_p, c_sz, c_p]),
    "mr_attn_bwd_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p, c_p]),
    "mr_attn_global_row_bwd_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p, c_p, c_p]),
    "mr_scatter_add_rows_f32": (c_i, [c_p, c_i64, c_p, c_i, c_i, c_p, c_i64, c_p]),
    "mr_split_tokens_kblock_f32": (c_i, [c_p, c_i64, c_i, c_i, c_i, c_p, c_p, c_p]),
    "mr_gemm_nt_bf16x3_splitk_ws_bytes": (c_sz, [c_i, c_i, c_i]),
    "mr_gemm_nt_bf16x3_splitk_f32": (c_i, [c_p, c_i64, c_p, c_p, c_i64, c_p, c_i, c_i, c_i, c_p, c_i64, c_p, c_i64, c_i, c_p, c_sz, c_p]),
    "mr_adamw_step_f32": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_i, c_d, c_d, c_d, c_d, c_d, c_i64, c_p, c_f, c_p]),
    "mr_pack_tokens": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mr_pack_tokens_checked": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "mr_embed_gather_ln_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p, c_f, c_i, c_i, c_i, c_p, c_p]),
    "mr_gemm_nt_bias_act_f32": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_p, c_i64, c_p]),
    "mr_gemm_nt_bf16x6_f32": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_i64, c_i64, c_i64, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_i64, c_p, c_i64, c_i, c_p]),
    "mr_split_weights_kblock_f16_f32": (c_i, [c_p, c_p, c_p, c_i, c_i64, c_p, c_p, c_p, c_p]),
    "mr_gemm_tile_f32": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_i64, c_p, c_p, c_p, c_i64, c_i, c_i,
                               c_p, c_p, c_p, c_i, c_p, c_i64, c_p, c_i64, c_f, c_u32, c_i, c_i, c_p]),
    "mr_distill_loss_rows_var_f32": (c_i, [c_p, c_i64, c_p, c_i64, c_i64, c_i64, c_p, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_p, c_p, c_i64, c_f, c_p]),
    "mr_skinny_scores_f32": (c_i, [c_p, c_i64, c_i, c_p, c_i64, c_i64, c_i, c_p, c_i64, c_p]),
    "mr_skinny_bwd_ws_bytes": (c_sz, [c_i, c_i64, c_i]),
    "mr_skinny_bwd_f32": (c_i, [c_p, c_i64, c_i, c_p, c_i64, c_i64, c_i, c_f, c_p, c_p, c_sz, c_p]),
    "mr_split_bf16x3_f32": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_p]),
    "mr_split_weights_kblock_f32": (c_i, [c_p, c_p, c_p, c_i, c_i64, c_p, c_p, c_p, c_p]),
    "mr_layernorm_f32": (c_i, [c_p, c_i64, c_p, c_p, c_f, c_i, c_i, c_p, c_i64, c_p]),
    "mr_attn_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p]),
    "mr_attn_split_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_p, c_p]),
    "mr_attn_split_q_rows": (c_i, [c_i, c_i]),
    "mr_attn_work_plan": (c_i64, [c_p, c_i, c_i, c_p, c_i64]),
    "mr_attn_work_f32": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_u32, c_p, c_p]),
    "mr_attn_split_work_f32": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_p, c_p]),
    "mr_dropout_site_key": (c_i, [c_u32, c_u32, c_u32, c_u32, c_p]),
    "mr_dropout_rows_f32": (c_i, [c_p, c_i64, c_i, c_i, c_f, c_u32, c_p, c_i64, c_p, c_i64, c_p]),
    "mr_attn_train_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_u32, c_p, c_p]),
    "mr_attn_split_work_train_f32": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_f, c_i, c_i, c_f, c_u32, c_p, c_p]),
    "mr_attn_global_row_train_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_f, c_u32, c_p, c_i, c_p]),
    "mr_attn_bwd_train_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_u32, c_p, c_p, c_p]),
    "mr_attn_bwd_work_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_f, c_i, c_f, c_u32, c_p, c_p, c_p]),
    "mr_attn_global_row_bwd_train_f32": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_u32, c_p, c_p, c_p]),
    "mr_attn_global_row_f32": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_p, c_i, c_p]),
    "mr_cls_pool_normalize_f32": (c_i, [c_p, c_i64, c_p, c_i, c_i, c_i, c_p, c_p]),
    "mr_mean_pool_f32": (c_i, [c_p, c_i64, c_p, c_p, c_p, c_i, c_i, c_i, c_p, c_p]),
    "mr_gather_rows_f32": (c_i, [c_p, c_i64, c_p, c_i, c_i, c_p, c_i64, c_p]),
    "mr_topk_rows_f32": (c_i, [c_p, c_i64, c_i, c_i, c_i, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p]),
    "mr_topk_max_k": (c_i, []),
    "mr_score_topk_ws_bytes": (c_sz, [c_i64, c_i64]),
    "mr_score_topk_ws_bytes_ex": (c_sz, [c_i64, c_i64, c_i, c_i]),
    "mr_score_fused_mode": (c_i, [c_i]),
    "mr_merge_bwd_generic": (c_i, [c_i]),
    "mr_score_topk_f32": (c_i, [c_p, c_p, c_i64, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_sz, c_p]),
}

_lib = None


def header_symbols() -> list[str]:
    """Every function name declared in include/mergerec_hip.h."""
    text = HEADER_PATH.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mr_[a-z0-9_]+)\s*\(", text)))


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("MERGEREC_HIP_LIB", LIB_PATH))  # override: A/B-testing another build of the same ABI
    if not path.exists():
        raise MergeRecHipError(
            f"{path} is missing: build it with `python -m mergerec_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the merged-inference path."
        )
    lib = ctypes.CDLL(str(path))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        lib = load()
        msg = lib.mr_strerror(rc).decode()
        hip = lib.mr_last_hip_error().decode()
        raise MergeRecHipError(f"{what} failed: {msg} (code {rc})" + (f"; HIP: {hip}" if hip else ""))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr(device=None):
    return torch.cuda.current_stream(device).cuda_stream
