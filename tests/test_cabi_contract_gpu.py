"""The C ABI's calling contract on the hot-path entry points (include/mergerec_hip.h:8-31), called through ctypes with raw device pointers:
zero-sized work is a successful no-op that leaves the outputs alone, bad arguments come back as MR_E* codes (never as a launch, a fault or
a C++ exception), a short workspace is MR_EWS, a misaligned pointer MR_EALIGN, and every code has a message."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MR_OK, MR_EINVAL, MR_EALIGN, MR_ELAUNCH, MR_EWS, MR_EUNSUPPORTED = 0, -1, -2, -3, -4, -5
SENTINEL = 7.25


@pytest.fixture(scope="module")
def lib():
    from mergerec_amd import _lib

    return _lib.load()


def p(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else None)


def filled(*shape, dtype=torch.float32):
    return torch.full(shape, SENTINEL, dtype=dtype, device=DEV)


def untouched(*ts):
    torch.cuda.synchronize()
    return all(bool((t == SENTINEL).all()) for t in ts)


def test_every_code_has_a_message(lib):
    seen = set()
    for code in (MR_OK, MR_EINVAL, MR_EALIGN, MR_ELAUNCH, MR_EWS, MR_EUNSUPPORTED):
        msg = lib.mr_strerror(code).decode()
        assert msg and msg not in seen, (code, msg)
        seen.add(msg)
    assert lib.mr_strerror(-99).decode()  # unknown codes still give a string
    assert lib.mr_version() > 0


def test_merge_contract(lib):
    P, N = 1024, 3
    base, tv, alpha, out = torch.randn(P, device=DEV), torch.randn(N, P, device=DEV), torch.rand(N, device=DEV), filled(P)
    call = lambda b=base, t=tv, a=alpha, o=out, n=N, s=1, seg=None, pb=0, pc=P, stride=P: lib.mr_merge_nway_f32(
        p(b), p(t), stride, p(a), p(seg), n, s, pb, pc, p(o), None)
    assert call(pc=0) == MR_OK and untouched(out), "an empty slice is a no-op"
    assert call(b=None) == MR_EINVAL and call(o=None) == MR_EINVAL and call(a=None) == MR_EINVAL
    assert call(n=0) == MR_EINVAL and call(pc=-4) == MR_EINVAL and call(pb=-4) == MR_EINVAL
    assert call(s=2, seg=None) == MR_EINVAL, "a segment table is required beyond one segment"
    assert call(pc=P - 2) == MR_EALIGN and call(pb=2, pc=P - 4) == MR_EALIGN and call(stride=P + 1) == MR_EALIGN
    assert call(b=base[1:], pc=P - 4) == MR_EALIGN, "base 4 bytes off a 16-byte boundary"
    assert untouched(out)
    assert call() == MR_OK
    torch.cuda.synchronize()
    assert torch.equal(out, base + (alpha[:, None] * tv).sum(0)) or torch.allclose(out, base + (alpha[:, None] * tv).sum(0), atol=1e-6)
    # task vectors
    tvo = filled(P)
    assert lib.mr_task_vector_f32(p(base), p(base), 0, p(tvo), None) == MR_OK and untouched(tvo)
    assert lib.mr_task_vector_f32(None, p(base), P, p(tvo), None) == MR_EINVAL
    assert lib.mr_task_vector_f32(p(base), p(base), -1, p(tvo), None) == MR_EINVAL
    assert lib.mr_task_vector_f32(p(base[1:]), p(base), P - 4, p(tvo), None) == MR_EALIGN and untouched(tvo)


def test_scoring_contract(lib):
    nU, M, d, k = 8, 300, 64, 50
    U, E = torch.randn(nU, d, device=DEV), torch.randn(M, d, device=DEV)
    val, idx = filled(nU, k), torch.full((nU, k), 77, dtype=torch.int64, device=DEV)
    prev = lib.mr_score_fused_mode(-1)
    try:
        for mode in (0, 1):  # staged route, fused route
            lib.mr_score_fused_mode(mode)
            need = lib.mr_score_topk_ws_bytes_ex(nU, M, d, k)
            assert 0 < need <= lib.mr_score_topk_ws_bytes(nU, M), "the shape-independent size is an upper bound"
            ws = torch.empty(need + 256, dtype=torch.uint8, device=DEV)
            call = lambda u=U, e=E, n=nU, m=M, dd=d, kk=k, v=val, i=idx, w=ws, wb=need: lib.mr_score_topk_f32(
                p(u), p(e), n, m, dd, kk, p(v), p(i), None, None, 1.0, None, None, None, p(w), wb, None)
            assert call(n=0) == MR_OK and untouched(val), f"mode {mode}: no users is a no-op"
            assert call(u=None) == MR_EINVAL and call(e=None) == MR_EINVAL and call(m=0) == MR_EINVAL and call(dd=0) == MR_EINVAL
            assert call(n=-1) == MR_EINVAL and call(w=None) == MR_EINVAL
            assert call(wb=need // 2) == MR_EWS, f"mode {mode}: half the workspace"
            kmax = lib.mr_topk_max_k()
            assert kmax == 1024
            assert call(kk=kmax + 1) in (MR_EUNSUPPORTED, MR_EWS) and call(kk=M + 1, m=M) in (MR_EUNSUPPORTED, MR_EWS), "k beyond the limit / beyond the catalog"
            assert untouched(val) and bool((idx == 77).all())
            assert call() == MR_OK
            torch.cuda.synchronize()
            want = torch.topk(U @ E.T, k, dim=1)
            assert torch.allclose(val, want.values, atol=1e-4) and torch.equal(idx, want.indices)
            val.fill_(SENTINEL); idx.fill_(77)
    finally:
        lib.mr_score_fused_mode(prev)
    # row select on its own
    sc = torch.randn(4, 100, device=DEV)
    tv_, ti_ = filled(4, 10), torch.zeros(4, 10, dtype=torch.int64, device=DEV)
    sel = lambda s=sc, ld=100, r=4, c=100, kk=10: lib.mr_topk_rows_f32(p(s), ld, r, c, kk, p(tv_), p(ti_), None, 1.0, None, None, None, None)
    assert sel(r=0) == MR_OK and untouched(tv_)
    assert sel(s=None) == MR_EINVAL and sel(r=-1) == MR_EINVAL and sel(c=0) == MR_EINVAL and sel(kk=0) == MR_EINVAL and sel(ld=99) == MR_EINVAL
    assert sel(kk=1025) == MR_EUNSUPPORTED and sel(kk=101) == MR_EUNSUPPORTED and sel(c=5, ld=5, kk=10) == MR_EUNSUPPORTED and untouched(tv_)


def test_encoder_kernels_contract(lib):
    T, d = 64, 768
    x, gam, bet, out = torch.randn(T, d, device=DEV), torch.ones(d, device=DEV), torch.zeros(d, device=DEV), filled(T, d)
    ln = lambda xx=x, t=T, dd=d, o=out, ldx=d, ldo=d: lib.mr_layernorm_f32(p(xx), ldx, p(gam), p(bet), 1e-5, t, dd, p(o), ldo, None)
    assert ln(t=0) == MR_OK and untouched(out)
    assert ln(xx=None) == MR_EINVAL and ln(t=-1) == MR_EINVAL and ln(dd=0) == MR_EINVAL
    assert ln(dd=770) == MR_EUNSUPPORTED and ln(dd=4096) == MR_EUNSUPPORTED
    assert ln(ldx=d + 1) == MR_EALIGN and ln(xx=x.view(-1)[1:], t=T - 1) == MR_EALIGN and untouched(out)
    assert ln() == MR_OK
    torch.cuda.synchronize()
    assert torch.allclose(out, torch.nn.functional.layer_norm(x, (d,), gam, bet, 1e-5), atol=1e-5)
    # exact-fp32 GEMM and the split-bf16 GEMM: no rows is a no-op, bad shapes are codes
    N, K = 256, 128
    A, W, C = torch.randn(T, K, device=DEV), torch.randn(N, K, device=DEV), filled(T, N)
    g32 = lambda a=A, w=W, c=C, m=T, n=N, kk=K, lda=K: lib.mr_gemm_nt_bias_act_f32(p(a), lda, p(w), None, None, None, None, None, 1, m, n, kk, 0,
                                                                                  None, 0, p(c), N, None)
    assert g32(m=0) == MR_OK and untouched(C)
    assert g32(a=None) == MR_EINVAL and g32(w=None) == MR_EINVAL and g32(c=None) == MR_EINVAL and g32(m=-1) == MR_EINVAL and g32(kk=0) == MR_EINVAL
    assert untouched(C)
    hi, mid = (torch.zeros(N * K, dtype=torch.int16, device=DEV) for _ in range(2))
    gsp = lambda a=A, m=T, kk=K, prod=3, act=0, lda=K: lib.mr_gemm_nt_bf16x6_f32(p(a), lda, p(hi), p(mid), None, 0, 0, 0, None, None, None, 1, m, N, kk,
                                                                               act, None, 0, p(C), N, prod, None)
    assert gsp(m=0) == MR_OK and untouched(C)
    assert gsp(a=None) == MR_EINVAL and gsp(m=-1) == MR_EINVAL
    assert gsp(prod=4) == MR_EUNSUPPORTED and gsp(act=9) == MR_EUNSUPPORTED and gsp(kk=K + 1) == MR_EUNSUPPORTED
    assert gsp(prod=6) == MR_EINVAL, "six products need the third piece"
    assert gsp(lda=K + 1) == MR_EALIGN and untouched(C)
    # attention: an empty batch is a no-op; only 64-wide heads exist
    H = 12
    qkv, ctx = torch.randn(T, 3 * d, device=DEV), filled(T, d)
    cu = torch.tensor([0, T], dtype=torch.int32, device=DEV)
    for fn in (lambda b, dh, q=qkv: lib.mr_attn_f32(p(q), p(cu), None, b, H, dh, T, 0.125, -1, p(ctx), None),
               lambda b, dh, q=qkv: lib.mr_attn_split_f32(p(q), p(cu), None, b, H, dh, T, 0.125, -1, 3, p(ctx), None)):
        assert fn(0, 64) == MR_OK and untouched(ctx)
        assert fn(-1, 64) == MR_EINVAL and fn(1, 32) == MR_EUNSUPPORTED and fn(1, 64, None) == MR_EINVAL and untouched(ctx)
