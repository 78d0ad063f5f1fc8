"""ctypes binding of oracle/oracle_c.c (TEST INFRASTRUCTURE ONLY -- see oracle/ref_cpu.py header)."""
import ctypes
import os
import subprocess
from pathlib import Path

import torch

_HERE = Path(__file__).resolve().parent
_LIB = _HERE / "_build" / "liboracle_c.so"


def build():
    src = _HERE / "oracle_c.c"
    if not _LIB.exists() or _LIB.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(["make", "-C", str(_HERE)], check=True, capture_output=True)
    return _LIB


def _lib():
    override = os.environ.get("ORACLE_C_LIB")  # a sanitizer build of the same source (oracle/Makefile: asan)
    lib = ctypes.CDLL(override if override else str(build()))
    return lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def gemm_nt(A: torch.Tensor, W: torch.Tensor, bias=None) -> torch.Tensor:
    """C[m, n] = fmaf-chain_k(A[m, k] * W[n, k]) (+ bias[n]); ascending k."""
    A = A.contiguous().float()
    W = W.contiguous().float()
    M, K = A.shape
    N = W.shape[0]
    C = torch.empty(M, N, dtype=torch.float32)
    b = bias.contiguous().float() if bias is not None else None
    _lib().gemm_nt_ref(_p(A), ctypes.c_int64(K), _p(W), _p(b), M, N, K, _p(C), ctypes.c_int64(N))
    return C


def merge_nway(base, tv, alpha, seg_off=None):
    base = base.contiguous().float()
    tv = tv.contiguous().float()
    alpha = alpha.contiguous().float().reshape(-1)
    N, P = tv.shape
    S = alpha.numel() // N
    out = torch.empty_like(base)
    so = seg_off.contiguous().to(torch.int64) if seg_off is not None else None
    _lib().merge_nway_ref(_p(base), _p(tv), ctypes.c_int64(P), _p(alpha), _p(so), N, S, ctypes.c_int64(P), _p(out))
    return out


def topk_rows(scores: torch.Tensor, k: int):
    scores = scores.contiguous().float()
    R, C = scores.shape
    val = torch.empty(R, k, dtype=torch.float32)
    idx = torch.empty(R, k, dtype=torch.int64)
    _lib().topk_rows_ref(_p(scores), ctypes.c_int64(C), R, C, k, _p(val), _p(idx))
    return val, idx
