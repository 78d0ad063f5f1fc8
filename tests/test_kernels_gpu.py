"""GPU parity: every HIP kernel, called through the C ABI, against the CPU oracle on seeded inputs."""
import math

import pytest
import torch

from oracle import c_oracle as CO
from oracle import ref_cpu as O

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from mergerec_amd import ops as _ops

    return _ops


def _g(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------ K1 / K1b / a3
@pytest.mark.parametrize("N", [1, 2, 3, 8, 11])
def test_merge_taskwise_bitexact(ops, N):
    P = 64 * 1531  # multiple of 64 (arena granule), not of the chunk size
    g = _g(N)
    base = torch.randn(P, generator=g) * 0.02
    tv = torch.randn(N, P, generator=g) * 1e-3
    alpha = torch.rand(N, generator=g) * 0.6 - 0.1
    want = O.merge_task_wise(base, tv, alpha)
    got = ops.merge_nway(base.to(DEV), tv.to(DEV), alpha.to(DEV)).cpu()
    assert torch.equal(got, want)
    assert torch.equal(CO.merge_nway(base, tv, alpha), want)


def test_merge_layerwise_segments_and_slices_bitexact(ops):
    g = _g(7)
    seg_len = [64 * 3, 64 * 40, 64 * 1, 64 * 129, 64 * 17, 64 * 64]
    seg_off = torch.tensor([0] + list(torch.tensor(seg_len).cumsum(0)), dtype=torch.int64)
    P, N, S = int(seg_off[-1]), 3, len(seg_len)
    base = torch.randn(P, generator=g)
    tv = torch.randn(N, P, generator=g)
    alpha = torch.rand(S, N, generator=g)
    want = torch.empty(P)
    for s in range(S):
        a, b = int(seg_off[s]), int(seg_off[s + 1])
        want[a:b] = O.merge_task_wise(base[a:b], tv[:, a:b], alpha[s])
    got = ops.merge_nway(base.to(DEV), tv.to(DEV), alpha.to(DEV), seg_off.to(DEV)).cpu()
    assert torch.equal(got, want)
    # rank-slice form: two ranks each merge half of the arena into the same output buffer
    out = torch.full((P,), float("nan"), device=DEV)
    half = (P // 2) // 64 * 64
    ops.merge_nway(base.to(DEV), tv.to(DEV), alpha.to(DEV), seg_off.to(DEV), out=out, p_begin=0, p_count=half)
    ops.merge_nway(base.to(DEV), tv.to(DEV), alpha.to(DEV), seg_off.to(DEV), out=out, p_begin=half, p_count=P - half)
    assert torch.equal(out.cpu(), want)


def test_task_vector_bitexact(ops):
    g = _g(3)
    a, b = torch.randn(64 * 77 + 3, generator=g), torch.randn(64 * 77 + 3, generator=g)
    assert torch.equal(ops.task_vector(a.to(DEV), b.to(DEV)).cpu(), a - b)


def test_merge_bwd_alpha(ops):
    g = _g(5)
    seg_len = [64 * 300, 64 * 7, 64 * 1000]
    seg_off = torch.tensor([0] + list(torch.tensor(seg_len).cumsum(0)), dtype=torch.int64)
    P, N = int(seg_off[-1]), 4
    tv, gr = torch.randn(N, P, generator=g), torch.randn(P, generator=g)
    got = ops.merge_bwd_alpha(tv.to(DEV), gr.to(DEV), seg_off.to(DEV)).cpu()
    got2 = ops.merge_bwd_alpha(tv.to(DEV), gr.to(DEV), seg_off.to(DEV)).cpu()
    assert torch.equal(got, got2), "two-stage reduction must be bitwise reproducible"
    for s in range(3):
        a, b = int(seg_off[s]), int(seg_off[s + 1])
        want = O.merge_bwd_alpha(tv[:, a:b], gr[a:b])
        assert torch.allclose(got[s], want, rtol=1e-4, atol=1e-2 * math.sqrt((b - a) / 1e6)), (got[s], want)
    one = ops.merge_bwd_alpha(tv.to(DEV), gr.to(DEV)).cpu()
    assert torch.allclose(one[0], O.merge_bwd_alpha(tv, gr), rtol=1e-4, atol=2e-2)
    # the single-pass kernel (N <= 8: every stream of a chunk in flight at once) keeps each sum's operation order: equal, bit for bit, to
    # the per-vector loop that larger N still use -- for every N, with segments whose last chunk is ragged and shorter than a chunk
    from mergerec_amd import _lib

    for n in (1, 2, 3, 5, 8):
        tvn = torch.randn(n, P, generator=g).to(DEV)
        fast = ops.merge_bwd_alpha(tvn, gr.to(DEV), seg_off.to(DEV)).cpu()
        assert _lib.load().mr_merge_bwd_generic(1) == 0
        try:
            slow = ops.merge_bwd_alpha(tvn, gr.to(DEV), seg_off.to(DEV)).cpu()
        finally:
            assert _lib.load().mr_merge_bwd_generic(0) == 1
        assert torch.equal(fast, slow), n


# ------------------------------------------------------------------ K3 GEMM (bit-exact vs k-ordered fmaf chain)
@pytest.mark.parametrize("M,N,K", [(1, 64, 64), (130, 768, 768), (257, 200, 3072), (64, 333, 16), (1024, 4096, 64), (1000, 4000, 32)])
def test_gemm_bitexact_vs_fma_chain(ops, M, N, K):
    # the plain product picks 64- or 128-row block tiles by how evenly they load the CUs (csrc/gemm.hip): the first four shapes and the
    # last take the half tile, (1024, 4096, 64) exactly one round of 128-row tiles -- the k chain per element is the same in both
    g = _g(M + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.05, torch.randn(N, generator=g)
    got = ops.gemm_nt(A.to(DEV), [W.to(DEV)], [b.to(DEV)]).cpu()
    want = CO.gemm_nt(A, W, b)
    assert torch.equal(got, want), (got - want).abs().max()
    assert torch.allclose(got, A @ W.T + b, atol=1e-4, rtol=1e-5)  # the reference's own arithmetic (torch GEMM)


def test_gemm_segments_gelu_residual(ops):
    g = _g(11)
    M, K, n = 200, 128, 128
    A = torch.randn(M, K, generator=g)
    Ws = [torch.randn(n, K, generator=g) * 0.1 for _ in range(3)]
    bs = [torch.randn(n, generator=g) for _ in range(3)]
    got = ops.gemm_nt(A.to(DEV), [w.to(DEV) for w in Ws], [b.to(DEV) for b in bs]).cpu()
    want = torch.cat([CO.gemm_nt(A, w, b) for w, b in zip(Ws, bs)], dim=1)
    assert torch.equal(got, want)
    R = torch.randn(M, n, generator=g)
    got = ops.gemm_nt(A.to(DEV), [Ws[0].to(DEV)], [bs[0].to(DEV)], act=ops.ACT_GELU, residual=R.to(DEV)).cpu()
    want = torch.nn.functional.gelu(CO.gemm_nt(A, Ws[0], bs[0])) + R
    assert torch.allclose(got, want, atol=2e-6, rtol=1e-6)


@pytest.mark.parametrize("M,N,K", [(130, 768, 768), (257, 200, 3072), (64, 333, 16), (300, 896, 64), (6656, 2560, 32)])
def test_gemm_bf16x6_split_precision(ops, M, N, K):
    """6 bf16 products per fp32 product: fp32-grade accuracy (not bit-exact); weights addressed inside split arenas.  The last two shapes
    put a narrower last group into the kernel's column-group tile order: 7 narrow column tiles = groups of 6 + 1, and (52 row tiles x 10
    wide column tiles >= 512 workgroups: the 128 x 256 tile kernel) groups of 3 + 3 + 3 + 1."""
    g = _g(M * 3 + N + K)
    A, b = torch.randn(M, K, generator=g), torch.randn(N, generator=g)
    arena = torch.zeros(64 * 5 + N * K + 64)  # the matrix sits at a non-zero, 64-aligned offset of a bigger arena
    off = 64 * 5
    W = torch.randn(N, K, generator=g) * 0.05
    arena[off : off + N * K] = W.reshape(-1)
    ad = arena.to(DEV)
    flat_pieces = ops.split_bf16x3(ad)
    hi, mid, lo = (p.float().cpu() for p in flat_pieces)
    assert torch.equal(hi, arena.to(torch.bfloat16).float())
    assert float((arena - (hi + mid + lo)).abs().max()) <= 2.0 ** -23 * float(arena.abs().max())
    pieces = ops.split_weights_kblock(ad, ops.KBlockTable([(off, N, K)], DEV))
    # k-blocked layout: element (n, k) at off + ((k // 16) * N + n) * 16 + k % 16
    kb = pieces[0].float().cpu()[off : off + N * K].view(K // 16, N, 16).permute(1, 0, 2).reshape(N, K)
    assert torch.equal(kb, W.to(torch.bfloat16).float())
    got = ops.gemm_nt_split(A.to(DEV), pieces, [off], N, K, [b.to(DEV)]).cpu()
    ref = (A.double() @ W.double().T + b.double())
    scale = (A.abs().double() @ W.abs().double().T).max()
    assert float((got.double() - ref).abs().max()) <= 4e-7 * float(scale), float((got.double() - ref).abs().max() / scale)
    R = torch.randn(M, N, generator=g)
    got = ops.gemm_nt_split(A.to(DEV), pieces, [off], N, K, [b.to(DEV)], act=ops.ACT_GELU, residual=R.to(DEV)).cpu()
    want = torch.nn.functional.gelu(ref.float()) + R
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("M,N,K", [(130, 768, 768), (257, 200, 3072), (64, 333, 16), (300, 896, 64), (6656, 2560, 32)])
def test_gemm_f16x3_split_precision(ops, M, N, K):
    """f16x3 (r04): two FP16 pieces per operand, three products.  The weight pieces hold 256 w (exact scale, undone in the epilogue) so that
    the low piece of a 0.01..0.1 weight is a normal fp16 number; activations are split unscaled, their low pieces may be fp16 subnormals,
    which the matrix pipe honours.  Error ~2^-21 per product: 30x tighter than bf16x3's bound at the same matrix-pipe cost; tiny and huge
    (but in-range) operands included."""
    g = _g(M * 5 + N + K)
    A, b = torch.randn(M, K, generator=g), torch.randn(N, generator=g)
    A[::7] *= 1e-3          # rows of small activations: low pieces in fp16's subnormal range
    A[1::11, ::13] *= 300.0  # massive activations
    arena = torch.zeros(64 * 5 + N * K + 64)
    off = 64 * 5
    W = torch.randn(N, K, generator=g) * 0.05
    W[::5] *= 1e-3
    arena[off : off + N * K] = W.reshape(-1)
    ad = arena.to(DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    pieces = ops.split_weights_kblock(ad, ops.KBlockTable([(off, N, K)], DEV), f16=True, overflow=flag)
    assert pieces[0].dtype == torch.float16 and pieces[2] is None and int(flag.item()) == 0
    kb = lambda p: p.float().cpu()[off : off + N * K].view(K // 16, N, 16).permute(1, 0, 2).reshape(N, K)
    hi, lo = kb(pieces[0]), kb(pieces[1])
    assert torch.equal(hi, (W * 256).to(torch.float16).float())                      # hi = fp16(256 w), round-to-nearest-even
    assert torch.equal(lo, (W * 256 - hi).to(torch.float16).float())                 # lo = fp16 of the exact remainder
    assert float(((hi + lo) / 256 - W).abs().max()) <= 2.0 ** -21 * float(W.abs().max())
    got = ops.gemm_nt_split(A.to(DEV), pieces, [off], N, K, [b.to(DEV)], products=ops.PRODUCTS_F16X3).cpu()
    ref = (A.double() @ W.double().T + b.double())
    scale = (A.abs().double() @ W.abs().double().T).max()
    err = float((got.double() - ref).abs().max() / scale)
    assert err <= 2.0 ** -20, err
    x3 = ops.gemm_nt_split(A.to(DEV), ops.split_weights_kblock(ad, ops.KBlockTable([(off, N, K)], DEV), n_pieces=2), [off], N, K, [b.to(DEV)], products=3).cpu()
    err3 = float((x3.double() - ref).abs().max() / scale)
    print(f"[{M}x{N}x{K}] max |err| / max sum|a||w|: f16x3 {err:.2e} (2^{math.log2(err + 1e-300):.1f}), bf16x3 {err3:.2e} (2^{math.log2(err3 + 1e-300):.1f})")
    R = torch.randn(M, N, generator=g)
    got = ops.gemm_nt_split(A.to(DEV), pieces, [off], N, K, [b.to(DEV)], act=ops.ACT_GELU, residual=R.to(DEV), products=ops.PRODUCTS_F16X3).cpu()
    want = torch.nn.functional.gelu(ref.float()) + R
    assert float((got - want).abs().max()) <= 2.0 ** -19 * float(scale) + 2e-6


def test_f16x3_weight_range_is_checked(ops):
    """a weight whose 256-fold value leaves fp16's range (|w| >= 255.9) or a NaN raises the split's overflow flag; in-range extremes do not"""
    n, K = 128, 64
    for val, bad in ((255.0, False), (256.0, True), (float("nan"), True), (-255.8, False), (-1e4, True)):
        arena = torch.zeros(n * K)
        arena[1234] = val
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.split_weights_kblock(arena.to(DEV), ops.KBlockTable([(0, n, K)], DEV), f16=True, overflow=flag)
        assert bool(flag.item()) == bad, val


def test_gemm_bf16x6_segments(ops):
    g = _g(77)
    M, K, n = 150, 128, 128
    A = torch.randn(M, K, generator=g)
    Ws = [torch.randn(n, K, generator=g) * 0.1 for _ in range(3)]
    bs = [torch.randn(n, generator=g) for _ in range(3)]
    arena = torch.cat([Ws[0].reshape(-1), torch.zeros(64), Ws[1].reshape(-1), torch.zeros(128), Ws[2].reshape(-1)])
    offs = [0, n * K + 64, 2 * n * K + 64 + 128]
    pieces = ops.split_weights_kblock(arena.to(DEV), ops.KBlockTable([(o, n, K) for o in offs], DEV))
    got = ops.gemm_nt_split(A.to(DEV), pieces, offs, n, K, [b.to(DEV) for b in bs]).cpu()
    want = torch.cat([A @ w.T + b for w, b in zip(Ws, bs)], dim=1)
    assert torch.allclose(got, want, atol=2e-5, rtol=1e-5)


# ------------------------------------------------------------------ K2 / LN / pooling
def _ragged_batch(B, L, vocab, g, pad=1):
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    ids = torch.randint(3, vocab, (B, L), generator=g)
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    ids[:, 0] = 0
    ids = torch.where(mask.bool(), ids, torch.full_like(ids, pad))
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = lens.cumsum(0)
    return ids, mask, lens, cu


def test_pack_and_embed_roberta(ops):
    g = _g(21)
    cfg = O.EncoderConfig(hidden=768, vocab=500, max_pos=140, layers=0)
    sd = O.random_state_dict(O.roberta_param_shapes(cfg, pooler=False), seed=5)
    ids, mask, lens, cu = _ragged_batch(9, 130, cfg.vocab, g)
    ids[3, 2] = cfg.pad_id  # a pad id inside a sequence: position ids must follow the cumsum rule
    T = int(cu[-1])
    tw, tp, _, _ = ops.pack_tokens(ids.to(DEV), mask.to(DEV), cu.to(DEV), T, cfg.pad_id)
    want_pos = O.position_ids_from_input_ids(ids, cfg.pad_id)[mask.bool()]
    assert torch.equal(tw.cpu().long(), ids[mask.bool()])
    assert torch.equal(tp.cpu().long(), want_pos)
    d = {k: v.to(DEV) for k, v in sd.items()}
    e = "model.embeddings."
    got = ops.embed_gather_ln(tw, tp, None, None, d[e + "word_embeddings.weight"], d[e + "position_embeddings.weight"],
                              d[e + "token_type_embeddings.weight"], None, d[e + "LayerNorm.weight"], d[e + "LayerNorm.bias"],
                              cfg.ln_eps, ops.EMBED_ROBERTA).cpu()
    want = O.roberta_embeddings(sd, ids, cfg, "model.")[mask.bool()]
    assert torch.allclose(got, want, atol=2e-6, rtol=1e-6), (got - want).abs().max()


def test_pack_and_embed_recformer(ops):
    g = _g(22)
    cfg = O.EncoderConfig(hidden=64, vocab=300, max_pos=200, layers=0, token_type_size=4, max_item_embeddings=51)
    sd = O.random_state_dict(O.recformer_param_shapes(cfg), seed=6, std=0.2)
    ids, mask, lens, cu = _ragged_batch(6, 77, cfg.vocab, g)
    tt = torch.randint(0, 4, ids.shape, generator=g)
    ip = torch.randint(0, 51, ids.shape, generator=g)
    T = int(cu[-1])
    tw, tp, ttp, tip = ops.pack_tokens(ids.to(DEV), mask.to(DEV), cu.to(DEV), T, cfg.pad_id, tt.to(DEV), ip.to(DEV))
    d = {k: v.to(DEV) for k, v in sd.items() if v.is_floating_point()}
    e = "model.embeddings."
    got = ops.embed_gather_ln(tw, tp, ttp, tip, d[e + "word_embeddings.weight"], d[e + "position_embeddings.weight"],
                              d[e + "token_type_embeddings.weight"], d[e + "item_position_embeddings.weight"],
                              d[e + "LayerNorm.weight"], d[e + "LayerNorm.bias"], cfg.ln_eps, ops.EMBED_RECFORMER).cpu()
    want = O.recformer_embeddings(sd, ids, tt, ip, cfg, "model.")[mask.bool()]
    assert torch.allclose(got, want, atol=5e-6, rtol=1e-6), (got - want).abs().max()


@pytest.mark.parametrize("d", [64, 768, 1024])
def test_layernorm_and_cls_pool(ops, d):
    g = _g(d)
    T = 301
    x, w, b = torch.randn(T, d, generator=g) * 3 + 1, torch.randn(d, generator=g), torch.randn(d, generator=g)
    got = ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5).cpu()
    assert torch.allclose(got, torch.nn.functional.layer_norm(x, (d,), w, b, 1e-5), atol=5e-6, rtol=1e-6)
    cu = torch.tensor([0, 5, 6, 100, 301], dtype=torch.int32)
    got = ops.cls_pool_normalize(x.to(DEV), cu.to(DEV), 4, True).cpu()
    assert torch.allclose(got, O.maybe_normalize(x[cu[:-1].long()]), atol=1e-6)
    got = ops.cls_pool_normalize(x.to(DEV), cu.to(DEV), 4, False).cpu()
    assert torch.equal(got, x[cu[:-1].long()])
    idx = torch.tensor([300, 0, 17], dtype=torch.int32)
    assert torch.equal(ops.gather_rows(x.to(DEV), idx.to(DEV)).cpu(), x[idx.long()])
    y = torch.randn(9, 333)  # unaligned width: scalar path
    assert torch.equal(ops.gather_rows(y.to(DEV), idx.to(DEV) % 9).cpu(), y[(idx % 9).long()])
    z = torch.randn(5, 4 * 2048 + 77)  # catalog-length rows (teacher scores): several column pieces per row, ragged last piece
    assert torch.equal(ops.gather_rows(z.to(DEV), idx.to(DEV) % 5).cpu(), z[(idx % 5).long()])


# ------------------------------------------------------------------ K4 attention
def _attn_ref(qkv, cu, H, window=-1):
    T, d3 = qkv.shape
    d = d3 // 3
    dh = d // H
    out = torch.zeros(T, d)
    for b in range(len(cu) - 1):
        a, e = int(cu[b]), int(cu[b + 1])
        L = e - a
        q, k, v = (qkv[a:e, i * d:(i + 1) * d].view(L, H, dh).transpose(0, 1) for i in range(3))
        s = (q @ k.transpose(-1, -2)) * dh ** -0.5
        if window >= 0:
            i = torch.arange(L)
            ok = ((i[:, None] - i[None, :]).abs() <= window) | (i[None, :] == 0)
            s = s.masked_fill(~ok[None], float("-inf"))
        out[a:e] = (torch.softmax(s, -1) @ v).transpose(0, 1).reshape(L, d)
    return out


SPLIT_UNIT = {3: 2.0 ** -15, 6: 2.0 ** -22, 35: 2.0 ** -20}   # per-product relative error of the split arithmetics (two / three bf16 pieces
                                               # per operand: 2^-17 / 2^-25 representation error per operand plus the dropped low x low products,
                                               # doubled; 35 = f16x3: two fp16 pieces, 2^-23 + 2^-22)


def _attn_split_bound(qkv, cu, H, window, products):
    """A-priori error bound of split-precision attention DERIVED FROM THE INPUTS (fp64 on the host), per output element:
    logit error  delta_i <= u * scale * max_j sum_k |q_ik| |k_jk|;  softmax under a logit perturbation of at most delta:
    |p~_ij - p_ij| <= p_ij * expm1(2 delta_i);  P V evaluated with per-product error u:
        |o~_id - o_id| <= (expm1(2 delta_i) + u) * sum_j p_ij |v_jd|   (+ fp32 accumulation: 64 ulp of the same sum).
    It scales with the logits' magnitude -- peaky (trained-like) rows get a wider bound than unit-variance ones -- instead of one
    absolute constant tuned on unit-variance inputs."""
    u = SPLIT_UNIT[products]
    qkv = qkv.double()
    T, d3 = qkv.shape
    d = d3 // 3
    dh = d // H
    bound = torch.zeros(T, d, dtype=torch.float64)
    for b in range(len(cu) - 1):
        a, e = int(cu[b]), int(cu[b + 1])
        L = e - a
        q, k, v = (qkv[a:e, i * d:(i + 1) * d].view(L, H, dh).transpose(0, 1) for i in range(3))
        s = (q @ k.transpose(-1, -2)) * dh ** -0.5
        mag = (q.abs() @ k.abs().transpose(-1, -2)) * dh ** -0.5
        if window >= 0:
            i = torch.arange(L)
            ok = ((i[:, None] - i[None, :]).abs() <= window) | (i[None, :] == 0)
            s = s.masked_fill(~ok[None], float("-inf"))
            mag = mag.masked_fill(~ok[None], 0.0)
        delta = u * mag.max(-1, keepdim=True).values
        abs_v = 0.0
        if products == 35:  # fp16 pieces: a low piece below 2^-14 is an fp16 subnormal, i.e. the value is held to an ABSOLUTE 2^-25
            delta = delta + dh * 2.0 ** -25 * (q.abs().amax(-1, keepdim=True) * dh ** -0.5 + k.abs().max())
            abs_v = 2.0 ** -25
        pv = torch.softmax(s, -1) @ v.abs()
        bound[a:e] = ((torch.expm1(2 * delta) + u + 64 * 2.0 ** -24) * pv + abs_v).transpose(0, 1).reshape(L, d)
    return bound


@pytest.mark.parametrize("H,lens", [(12, [512, 1, 33, 40, 257]), (4, [5, 64, 31, 32, 96])])
def test_attention_full(ops, H, lens):
    g = _g(H)
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    T = int(cu[-1])
    qkv = torch.randn(T, 3 * H * 64, generator=g)
    got = ops.attention(qkv.to(DEV), cu.to(DEV), len(lens), H, max(lens)).cpu()
    want = _attn_ref(qkv, cu, H)
    assert torch.allclose(got, want, atol=3e-6, rtol=1e-5), (got - want).abs().max()


@pytest.mark.parametrize("products", [3, 6, 35])
@pytest.mark.parametrize("window", [-1, 4, 32])
@pytest.mark.parametrize("qk_scale", [1.0, 2.5])   # 2.5: pre-softmax logits with sigma ~ 6 and |max| ~ 35 -- peaky rows, the rescale path taken
def test_attention_split_precision(ops, products, window, qk_scale):
    """Split-precision attention against fp64, inside the bound DERIVED from the inputs (``_attn_split_bound``), for unit-variance and
    for peaky (trained-like) logits; the observed error / bound ratio is printed."""
    g = _g(100 + products + window)
    H, lens = 4, [300, 1, 2, 70, 33, 129, 512]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    T = int(cu[-1])
    qkv = torch.randn(T, 3 * H * 64, generator=g)
    qkv[:, : 2 * H * 64] *= qk_scale
    ctx = torch.full((T, H * 64), 7.5, device=DEV)
    ops.attention(qkv.to(DEV), cu.to(DEV), len(lens), H, max(lens), window=window, out=ctx, products=products)
    got, want = ctx.cpu().double(), _attn_ref(qkv.double(), cu, H, window)
    bound = _attn_split_bound(qkv, cu, H, window, products)
    keep = torch.ones(T, dtype=torch.bool)
    if window >= 0:
        keep[cu[:-1].long()] = False
        assert bool((ctx.cpu()[~keep] == 7.5).all())
    err = (got - want).abs()[keep]
    assert bool((err <= bound[keep]).all()), float((err / bound[keep]).max())
    print(f"[products {products} window {window} qk x{qk_scale}] max |err| {float(err.max()):.2e}; bound max {float(bound[keep].max()):.2e}; worst err / bound {float((err / bound[keep]).max()):.3f}")


@pytest.mark.parametrize("products", [0, 3, 6, 35])
@pytest.mark.parametrize("window", [-1, 4, 32])
def test_attention_work_list_is_bit_identical_to_the_box_grid(ops, products, window):
    """mr_attn_work_f32 (products = 0: the exact-fp32 kernel) / mr_attn_split_work_f32 (host-built (sequence, query block) list, 256-row blocks with two query tiles per wave for full attention in
    bf16x3) against mr_attn_split_f32 on a ragged batch that exercises every tile-count case of a block (1..8 query tiles, partial
    tiles, one-token sequences): the same bits, rows that belong to the global-row kernel untouched, and within tolerance of fp64."""
    g = _g(300 + products + window)
    H, lens = 3, [512, 1, 2, 31, 32, 33, 64, 65, 127, 128, 129, 160, 191, 224, 255, 256, 257, 300, 384, 385, 511, 700, 1024]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    T = int(cu[-1])
    qkv = torch.randn(T, 3 * H * 64, generator=g).to(DEV)
    lens_t = torch.tensor(lens)
    work = {}
    for q in (128, 256):
        w, n = ops.attn_work_plan(lens_t, q)
        ents = w[w >= 0]
        cnt = torch.bincount(ents & 0xFFFFFF, minlength=len(lens))
        assert torch.equal(cnt, (lens_t + q - 1) // q) and w.numel() == 8 * n  # every (sequence, block) exactly once
        work[q] = (w.to(DEV), n)
    box = torch.full((T, H * 64), 7.5, device=DEV)
    lst = torch.full((T, H * 64), 7.5, device=DEV)
    ops.attention(qkv, cu.to(DEV), len(lens), H, max(lens), window=window, out=box, products=products)
    ops.attention(qkv, cu.to(DEV), len(lens), H, max(lens), window=window, out=lst, products=products, work=work)
    assert torch.equal(box, lst), (box - lst).abs().max()
    want = _attn_ref(qkv.cpu().double(), cu, H, window).float()
    keep = torch.ones(T, dtype=torch.bool)
    if window >= 0:
        keep[cu[:-1].long()] = False
        assert bool((lst.cpu()[~keep] == 7.5).all())
    if products:
        assert bool(((lst.cpu().double() - _attn_ref(qkv.cpu().double(), cu, H, window)).abs()[keep] <= _attn_split_bound(qkv.cpu(), cu, H, window, products)[keep]).all())
    else:
        assert torch.allclose(lst.cpu()[keep], want[keep], atol=5e-6, rtol=5e-6)


def test_attention_work_plan_rejects_bad_arguments(ops):
    from mergerec_amd._lib import MergeRecHipError

    with pytest.raises(MergeRecHipError):
        ops.attn_work_plan(torch.tensor([5, -1]), 256)
    with pytest.raises(MergeRecHipError):
        ops.attn_work_plan(torch.tensor([5]), 64)
    w, n = ops.attn_work_plan(torch.zeros(0, dtype=torch.int64), 256)
    assert n == 0 and w.numel() == 0


def test_work_list_entry_points_check_the_block_height_and_the_batch(ops):
    """An entry of a work list is a BLOCK index: the list only means something for the block height it was planned with and for its own
    batch.  The entry points take (B, q_rows) and refuse a mismatched height; a list planned for a larger batch cannot index cu_seqlens
    past its B + 1 entries (the kernel skips entries naming a sequence >= B)."""
    import ctypes

    from mergerec_amd import _lib

    lib, H = _lib.load(), 2
    lens = torch.tensor([300, 40, 257])
    cu = torch.tensor([0, 300, 340, 597], dtype=torch.int32, device=DEV)
    qkv = torch.randn(597, 3 * H * 64, generator=_g(9)).to(DEV)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    plans = {q: ops.attn_work_plan(lens, q) for q in (128, 256)}
    dev = {q: (w.to(DEV), n) for q, (w, n) in plans.items()}
    out = torch.full((597, H * 64), 7.5, device=DEV)
    # bf16x3 full attention owns 256 rows per entry, every other kernel 128
    assert lib.mr_attn_split_work_f32(P(qkv), P(cu), P(dev[128][0]), dev[128][1], 3, 128, H, 64, 0.125, -1, 3, P(out), None) == -1
    assert lib.mr_attn_split_work_f32(P(qkv), P(cu), P(dev[256][0]), dev[256][1], 3, 256, H, 64, 0.125, 4, 3, P(out), None) == -1
    assert lib.mr_attn_split_work_f32(P(qkv), P(cu), P(dev[256][0]), dev[256][1], 3, 256, H, 64, 0.125, -1, 6, P(out), None) == -1
    assert lib.mr_attn_work_f32(P(qkv), P(cu), P(dev[256][0]), dev[256][1], 3, 256, H, 64, 0.125, -1, 0.0, 0, P(out), None) == -1
    assert lib.mr_attn_split_work_f32(P(qkv), P(cu), P(dev[256][0]), dev[256][1], -1, 256, H, 64, 0.125, -1, 3, P(out), None) == -1
    torch.cuda.synchronize()
    assert bool((out == 7.5).all()), "a refused call must not launch"
    # a list planned for 3 sequences used with the first two only: rows of sequences 0 and 1 are computed, nothing past cu[2] is read or written
    want = torch.full_like(out, 7.5)
    ops.attention(qkv[:340], cu[:3], 2, H, 300, out=want[:340], products=3)
    assert lib.mr_attn_split_work_f32(P(qkv), P(cu), P(dev[256][0]), dev[256][1], 2, 256, H, 64, 0.125, -1, 3, P(out), None) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, want)


@pytest.mark.parametrize("window", [4, 32])
def test_attention_band_global(ops, window):
    g = _g(window)
    H, lens = 4, [300, 1, 2, 70, 33, 129]
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    T = int(cu[-1])
    qkv = torch.randn(T, 3 * H * 64, generator=g)
    sentinel = 1234.5
    ctx = torch.full((T, H * 64), sentinel, device=DEV)
    ops.attention(qkv.to(DEV), cu.to(DEV), len(lens), H, max(lens), window=window, out=ctx)
    got = ctx.cpu()
    want = _attn_ref(qkv, cu, H, window)
    first = torch.zeros(T, dtype=torch.bool)
    first[cu[:-1].long()] = True
    assert torch.allclose(got[~first], want[~first], atol=3e-6, rtol=1e-5), (got[~first] - want[~first]).abs().max()
    assert bool((got[first] == sentinel).all()), "row 0 of each sequence belongs to the global-row kernel"
    # global row
    qg = torch.randn(len(lens), H * 64, generator=g)
    kvg = torch.randn(T, 2 * H * 64, generator=g)
    ops.attention_global_row(qg.to(DEV), kvg.to(DEV), cu.to(DEV), len(lens), H, max(lens), ctx)
    got = ctx.cpu()
    for b in range(len(lens)):
        a, e = int(cu[b]), int(cu[b + 1])
        q = qg[b].view(H, 1, 64)
        k = kvg[a:e, : H * 64].view(-1, H, 64).transpose(0, 1)
        v = kvg[a:e, H * 64:].view(-1, H, 64).transpose(0, 1)
        w = (torch.softmax((q @ k.transpose(-1, -2)) * 0.125, -1) @ v).reshape(-1)
        assert torch.allclose(got[a], w, atol=3e-6, rtol=1e-5)


# ------------------------------------------------------------------ K5 scoring + top-k
def test_topk_rows_canonical_with_ties(ops):
    g = _g(31)
    R, C, k = 70, 4968, 50
    s = torch.randn(R, C, generator=g)
    s[0, torch.randperm(C, generator=g)[:200]] = float(s[0].max()) + 1.0   # > k equal maxima
    s[1, :] = 0.25                                                          # constant row
    s[2, 5] = float("nan"); s[2, 4000] = float("nan")
    s[3, 100:130] = float(s[3].topk(50).values[-1])                         # ties straddling the k-th value
    s[4, 7] = float("inf"); s[4, 9] = float("-inf")
    s[5, 10] = 0.0; s[5, 11] = -0.0; s[5, 12:] = -1.0; s[5, :10] = -2.0    # +0 == -0 tie by index
    labels = torch.randint(0, C, (R,), generator=g)
    labels[0] = int(torch.nonzero(s[0] == s[0].max())[3])
    val, idx, lse, lab, rank = ops.topk_rows(s.to(DEV), k, labels.to(DEV), 20.0)
    wv, wi = CO.topk_rows(s, k)
    assert torch.equal(idx.cpu(), wi)
    assert torch.equal(val.cpu().nan_to_num(7.0), wv.nan_to_num(7.0))
    ov, oi = O.topk_canonical(s, k)
    assert torch.equal(wi, oi)
    want_rank = torch.tensor([(wi[r] == labels[r]).nonzero()[0, 0].item() if (wi[r] == labels[r]).any() else -1 for r in range(R)], dtype=torch.int32)
    assert torch.equal(rank.cpu(), want_rank)
    ok = ~torch.isnan(s).any(1) & ~torch.isinf(s).any(1)
    want_lse = torch.logsumexp(s * 20.0, dim=1)
    assert torch.allclose(lse.cpu()[ok], want_lse[ok], atol=1e-4, rtol=1e-5)
    assert torch.allclose(lab.cpu()[ok], (s[torch.arange(R), labels] * 20.0)[ok], atol=1e-6)


@pytest.mark.parametrize("C,k", [(22855, 50), (22855, 64), (8191, 65), (8193, 200), (4968, 1000), (27932, 1024), (49152, 300), (1030, 1024),
                                 (50000, 50), (300, 7)])
def test_topk_rows_register_kernel_every_width_and_k(ops, C, k):
    """The register-resident row select (csrc/score.hip topk_rows_reg_kernel: every float4-count variant, k <= 64 sorted by one wavefront,
    larger k by the workgroup) and the LDS / L2 kernel behind it (C = 50,000) against the C oracle's canonical order: ties that straddle
    the k-th value, more equal maxima than k, NaN, +-inf, +-0, constant rows, an unaligned leading dimension."""
    g = _g(C + k)
    R = 9
    s = torch.randn(R, C, generator=g)
    s[0, torch.randperm(C, generator=g)[: k + 37]] = float(s[0].max()) + 1.0       # more equal maxima than k: lowest indices win
    s[1, :] = 0.25
    s[2, 5] = float("nan"); s[2, C - 1] = float("nan")
    kth = float(s[3].topk(k).values[-1])
    s[3, C // 2 : C // 2 + 40] = kth                                               # ties straddling the k-th value
    s[4, 7] = float("inf"); s[4, 9] = float("-inf")
    s[5, 10] = 0.0; s[5, 11] = -0.0; s[5, 12:] = -1.0; s[5, :10] = -2.0
    s[6] = torch.round(s[6] * 4) / 4                                               # heavy ties everywhere
    s[7] = -torch.rand(C, generator=g) * 1e-30                                     # tiny negatives: top byte of the key shared with nothing else
    labels = torch.randint(0, C, (R,), generator=g)
    labels[0] = int(torch.nonzero(s[0] == s[0].max())[3])
    wv, wi = CO.topk_rows(s, k)
    want_rank = torch.tensor([(wi[r] == labels[r]).nonzero()[0, 0].item() if (wi[r] == labels[r]).any() else -1 for r in range(R)], dtype=torch.int32)
    for pad in (0, 3):  # pad 3: leading dimension not a multiple of 4 -> the scalar load path
        buf = torch.full((R, C + pad), 9e9)
        buf[:, :C] = s
        val, idx, lse, lab, rank = ops.topk_rows(buf.to(DEV)[:, :C], k, labels.to(DEV), 20.0)
        assert torch.equal(idx.cpu(), wi), (pad, (idx.cpu() != wi).nonzero()[:5])
        assert torch.equal(val.cpu().nan_to_num(7.0), wv.nan_to_num(7.0))
        assert torch.equal(val.cpu()[5].view(torch.int32), wv[5].view(torch.int32))   # -0.0 keeps its sign bit
        assert torch.equal(rank.cpu(), want_rank)
        ok = ~torch.isnan(s).any(1) & ~torch.isinf(s).any(1)
        assert torch.allclose(lse.cpu()[ok], torch.logsumexp(s * 20.0, dim=1)[ok], atol=1e-4, rtol=1e-5)
        assert bool(torch.isnan(lse.cpu()[2]))
        assert torch.equal(lab.cpu()[ok], (s[torch.arange(R), labels] * 20.0)[ok])


def test_topk_rows_rejects_k_beyond_the_limit(ops):
    from mergerec_amd._lib import MergeRecHipError

    s = torch.randn(2, 3000).to(DEV)
    with pytest.raises(MergeRecHipError):
        ops.topk_rows(s, 1025)
    with pytest.raises(MergeRecHipError):
        ops.topk_rows(torch.randn(2, 50001).to(DEV), 65)   # rows beyond the register kernel: k <= 64
    with pytest.raises(MergeRecHipError):
        ops.topk_rows(torch.randn(2, 40).to(DEV), 41)      # k > ncols


@pytest.mark.parametrize("nU,M", [(33, 4968), (128, 18357), (5, 50)])
def test_score_topk_full_catalog(ops, nU, M):
    g = _g(M)
    d, k = 768, 50
    U = O.maybe_normalize(torch.randn(nU, d, generator=g))
    E = O.maybe_normalize(torch.randn(M, d, generator=g))
    E[M // 2] = E[3]  # duplicated catalog text -> identical embedding -> exact score tie
    labels = torch.randint(0, M, (nU,), generator=g)
    val, idx, lse, lab, rank, scores = ops.score_topk(U.to(DEV), E.to(DEV), k, labels.to(DEV), 20.0, return_scores=True)
    ref = O.score(U, E)                       # the reference's arithmetic: torch GEMM
    assert torch.allclose(scores.cpu(), ref, atol=1e-4, rtol=0), (scores.cpu() - ref).abs().max()   # north_star: logits within 1e-4
    chain = CO.gemm_nt(U, E)                  # same k-ordered fp32 FMA chain the MFMA path computes
    assert torch.equal(scores.cpu(), chain)
    wv, wi = CO.topk_rows(chain, k)
    assert torch.equal(idx.cpu(), wi), "ranked item indices must be bit-exact"
    assert torch.equal(val.cpu(), wv)
    _, oi = O.topk_canonical(ref, k)
    assert O.ranks_equal_up_to_ties(ref, idx.cpu(), oi, atol=2e-6)
    loss = float((lse - lab).mean())
    assert abs(loss - O.ce_loss(ref, labels, 0.05)) < 1e-3


def _assert_fused_equals_unfused(ops, U, E, k, labels, inv_temp=20.0):
    """score_topk without the score block (selection inside the scoring kernel, csrc/score_fused.hip) against the scoring GEMM + topk_rows
    pair: indices, values, label ranks and label logits bit for bit; the log-sum-exp to rounding (different summation order)."""
    Ud, Ed, Ld = U.to(DEV), E.to(DEV), labels.to(DEV)
    fv, fi, flse, flab, frank, none = ops.score_topk(Ud, Ed, k, Ld, inv_temp, fused=True)
    uv, ui, ulse, ulab, urank, scores = ops.score_topk(Ud, Ed, k, Ld, inv_temp, return_scores=True)
    assert none is None and scores is not None
    assert torch.equal(fi, ui), "fused ranked indices differ"
    assert torch.equal(fv.view(torch.int32), uv.view(torch.int32)), "fused top values differ (bit pattern, so NaN and -0.0 count)"
    assert torch.equal(frank, urank)
    assert torch.equal(flab.view(torch.int32), ulab.view(torch.int32))
    both_nan = torch.isnan(flse) & torch.isnan(ulse)
    assert torch.equal(torch.isnan(flse), torch.isnan(ulse))
    assert torch.allclose(flse[~both_nan], ulse[~both_nan], atol=2e-5, rtol=1e-6)
    return scores.cpu(), fi.cpu()


@pytest.mark.parametrize("nU,M,d,k", [(70, 4968, 768, 50), (256, 22855, 768, 50), (33, 768, 64, 50), (5, 769, 64, 64), (31, 50, 32, 50),
                                      (40, 1537, 48, 1)])
def test_score_topk_fused_matches_unfused(ops, nU, M, d, k):
    g = _g(nU * 7 + M)
    U = O.maybe_normalize(torch.randn(nU, d, generator=g))
    E = O.maybe_normalize(torch.randn(M, d, generator=g))
    if M > 1000:
        E[M // 2] = E[3]                      # an exact tie across two parts of the catalog
        E[5] = E[4]                           # and inside one part
    labels = torch.randint(0, M, (nU,), generator=g)
    labels[0] = M + 5                         # out of range: label logit NaN, rank -1
    scores, idx = _assert_fused_equals_unfused(ops, U, E, k, labels)
    _, wi = CO.topk_rows(scores, k)
    assert torch.equal(idx, wi)


def test_score_topk_auto_route_by_block_size(ops):
    """mode auto (the default): a (users x M) block beyond 128 MB could not stay in the Infinity Cache -> the fused kernels (workspace =
    candidate lists); a small block -> scoring GEMM + row select.  Same answers either way."""
    lib = ops._lib.load()
    assert lib.mr_score_fused_mode(-1) == 2
    g = _g(5)
    d, k = 64, 50
    for nU, M, want_fused in ((1500, 22855, True), (64, 4968, False)):
        assert (lib.mr_score_topk_ws_bytes_ex(nU, M, d, k) < 4 * nU * M) == want_fused
        U, E = torch.randn(nU, d, generator=g).to(DEV), torch.randn(M, d, generator=g).to(DEV)
        labels = torch.randint(0, M, (nU,), generator=g).to(DEV)
        av, ai, alse, alab, arank, _ = ops.score_topk(U, E, k, labels, 1.0)
        sv, si, slse, slab, srank, _ = ops.score_topk(U, E, k, labels, 1.0, return_scores=True)
        assert torch.equal(ai, si) and torch.equal(av, sv) and torch.equal(arank, srank) and torch.equal(alab, slab)
        assert torch.allclose(alse, slse, atol=2e-5, rtol=1e-6)


def test_score_topk_fused_ties_nan_and_signed_zero(ops):
    g = _g(99)
    nU, M, d, k = 40, 3000, 32, 50
    U = torch.randn(nU, d, generator=g)
    E = torch.randn(M, d, generator=g)
    E[100:400] = E[7]                          # 301 identical items: ties straddling the k-th value, by ascending index
    E[2000:2100] = E[7]                        # ... continuing in another part
    U[1] = 0.0                                 # every score +0.0 or -0.0 (E has both signs): one key, order by index
    U[2, 3] = float("nan")                     # every score NaN
    E[1234, 0] = float("nan")                  # one NaN column for everyone: ranks first
    U[3] = 1e30; E[50] = 1e30                  # +inf scores
    labels = torch.randint(0, M, (nU,), generator=g)
    labels[4] = 1234
    _assert_fused_equals_unfused(ops, U, E, k, labels, inv_temp=1.0)


@pytest.mark.parametrize("tool,env", [("gemm_fuzz.py", {"FZ_N": "25", "FZ_SEED": "3"}), ("attn_fuzz.py", {"FZ_N": "12", "FZ_SEED": "3"})])
def test_randomised_differential_fuzz(tool, env):
    """tools/gemm_fuzz.py / tools/attn_fuzz.py: random shapes, strides, epilogues, ragged lengths and windows against float64 torch"""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / tool)], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------ token-sized training products (csrc/gemm_train.hip)
@pytest.mark.parametrize("bn", [32, 64])
@pytest.mark.parametrize("ta,tb,M,N,K", [(False, False, 602, 768, 768), (False, False, 70, 192, 3072), (False, True, 602, 768, 2304), (False, True, 33, 3072, 768),
                                        (True, True, 768, 768, 602), (True, True, 2304, 768, 37), (True, True, 128, 3072, 1), (True, False, 64, 100, 32)])
def test_gemm_tile_is_the_k_ordered_fma_chain_in_every_orientation(ops, bn, ta, tb, M, N, K):
    """C = Aop Bop^T read from either orientation of either operand, 64 x 32 and 64 x 64 tiles on the two fp32 MFMA shapes: bit-equal to the
    C oracle's ascending-k FMA chain (oracle/oracle_c.c gemm_nt_ref), ragged M / N / K tails, token-deep K that is no multiple of 16."""
    g = _g(M + 3 * N + 7 * K + bn)
    Aop, Bop = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    A = (Aop.t().contiguous() if ta else Aop).to(DEV)
    B = (Bop.t().contiguous() if tb else Bop).to(DEV)
    got = ops.gemm_tile(A, [B], trans_a=ta, trans_b=tb, bn=bn).cpu()
    assert torch.equal(got, CO.gemm_nt(Aop, Bop)), float((got - CO.gemm_nt(Aop, Bop)).abs().max())


def test_gemm_tile_segments_and_epilogues(ops):
    g = _g(4242)
    T, d, di = 150, 128, 256
    x = torch.randn(T, d, generator=g)
    Ws = [torch.randn(d, d, generator=g) * 0.1 for _ in range(3)]
    bs = [torch.randn(d, generator=g) for _ in range(3)]
    xd = x.to(DEV)
    # forward of the stacked q / k / v projection: three weights, one launch
    qkv = ops.gemm_tile(xd, [w.to(DEV) for w in Ws], biases=[b.to(DEV) for b in bs]).cpu()
    want = torch.cat([CO.gemm_nt(x, w) + b for w, b in zip(Ws, bs)], dim=1)
    assert torch.equal(qkv, want)
    # its input gradient: dX = dQKV [Wq; Wk; Wv] (+ residual), the weights stacked along k
    dqkv, res = torch.randn(T, 3 * d, generator=g), torch.randn(T, d, generator=g)
    dx = ops.gemm_tile(dqkv.to(DEV), [w.to(DEV) for w in Ws], trans_b=True, residual=res.to(DEV)).cpu()
    wcat = torch.cat(Ws, dim=0)
    assert torch.equal(dx, CO.gemm_nt(dqkv, wcat.t().contiguous()) + res)
    # its weight gradients and bias gradients: three (d, d) outputs + three (d,) column sums from one (T, 3 d) dY
    dws = [torch.full((d, d), 7.5, device=DEV) for _ in range(3)]
    dbs = [torch.full((d,), 7.5, device=DEV) for _ in range(3)]
    ops.gemm_tile(dqkv.to(DEV), [xd], trans_a=True, trans_b=True, out=dws, colsum=dbs)
    for s_ in range(3):
        dy = dqkv[:, s_ * d:(s_ + 1) * d]
        assert torch.equal(dws[s_].cpu(), CO.gemm_nt(dy.t().contiguous(), x.t().contiguous()))
        assert torch.allclose(dbs[s_].cpu(), dy.sum(0), rtol=1e-5, atol=1e-5)
    # GELU forward (both outputs) and backward epilogues, dropout + residual
    W1, b1 = torch.randn(di, d, generator=g) * 0.1, torch.randn(di, generator=g)
    act = torch.empty(T, di, device=DEV)
    u = ops.gemm_tile(xd, [W1.to(DEV)], biases=[b1.to(DEV)], epi=ops.EPI_GELU_FWD, out2=act).cpu()
    assert torch.equal(u, CO.gemm_nt(x, W1) + b1)
    assert torch.allclose(act.cpu(), torch.nn.functional.gelu(u), atol=2e-6, rtol=1e-6)
    dyo = torch.randn(T, d, generator=g)
    W2 = torch.randn(d, di, generator=g) * 0.1
    du = ops.gemm_tile(dyo.to(DEV), [W2.to(DEV)], trans_b=True, epi=ops.EPI_GELU_BWD, E=u.to(DEV)).cpu()
    di_ = CO.gemm_nt(dyo, W2.t().contiguous())
    assert torch.allclose(du, ops.gelu_bwd(u.to(DEV), di_.to(DEV)).cpu(), atol=1e-6, rtol=1e-6)
    key = ops.dropout_site_key(3, 1, 0, ops.DROP_SITE_ATTN_OUT)
    y = ops.gemm_tile(xd, [Ws[0].to(DEV)], biases=[bs[0].to(DEV)], residual=res.to(DEV), drop_p=0.1, drop_key=key).cpu()
    plain = (CO.gemm_nt(x, Ws[0]) + bs[0]).to(DEV)
    assert torch.equal(y, ops.dropout_rows(plain, 0.1, key, residual=res.to(DEV)).cpu())


@pytest.mark.parametrize("bn", [32, 64])
@pytest.mark.parametrize("ta,tb,M,N,K", [(False, False, 602, 768, 768), (False, True, 602, 768, 2304), (True, True, 768, 768, 602), (True, True, 2304, 768, 37),
                                        (False, False, 70, 192, 3072), (True, False, 64, 100, 32)])
def test_gemm_tile_bf16x6_is_fp32_grade_over_the_whole_range(ops, bn, ta, tb, M, N, K):
    """products = 6: three bf16 pieces per operand split while the tile is staged, six MFMA products -- ~2^-24 per product for operands of
    ANY magnitude (rows of 1e-6-sized gradients beside O(1) activations: fp16 pieces would flush them), every orientation and tile width;
    the bias gradient that rides along (trans_a) against float64."""
    g = _g(M + 3 * N + 7 * K + bn + 1)
    Aop, Bop = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    Aop[::3] *= 1e-6
    Bop[1::4] *= 3e3
    A = (Aop.t().contiguous() if ta else Aop).to(DEV)
    B = (Bop.t().contiguous() if tb else Bop).to(DEV)
    cs = torch.full((M,), 7.5, device=DEV) if (ta and tb) else None
    got = ops.gemm_tile(A, [B], trans_a=ta, trans_b=tb, bn=bn, products=6, colsum=[cs] if cs is not None else None).cpu().double()
    want = Aop.double() @ Bop.double().T
    scale = Aop.abs().double() @ Bop.abs().double().T
    err = float(((got - want).abs() / scale).max())
    assert err <= 2.0 ** -21, err    # per element: relative to sum_k |a||b| (fp32 accumulation over K terms included)
    if cs is not None:
        assert torch.allclose(cs.cpu().double(), Aop.double().sum(1), rtol=1e-5, atol=1e-5 * float(Aop.abs().sum(1).max()))


def test_merge_rows_writes_the_listed_rows_with_the_bits_of_the_whole_merge(ops):
    """mr_merge_rows_f32: the rows of one table inside the arena, element for element the operations of mr_merge_nway_f32 (bit-identical),
    duplicates in the list harmless, every other element of the output untouched, out-of-table ids skipped"""
    g = torch.Generator().manual_seed(5)
    N, V, d, off = 8, 300, 64, 128
    P = off + V * d + 256
    base, tv = torch.randn(P, generator=g), torch.randn(N, P, generator=g)
    alpha = torch.randn(N, generator=g)
    whole = ops.merge_nway(base.to(DEV), tv.to(DEV), alpha.to(DEV))
    idx = torch.tensor([7, 0, 299, 7, 150, 151, 7, 42], dtype=torch.int32)
    out = torch.full((P,), 123.0, device=DEV)
    ops.merge_rows(base.to(DEV), tv.to(DEV), alpha.to(DEV), idx.to(DEV), V, d, off, out)
    touched = torch.zeros(P, dtype=torch.bool)
    for r in idx.tolist():
        touched[off + r * d: off + (r + 1) * d] = True
    assert torch.equal(out.cpu()[touched], whole.cpu()[touched])
    assert bool((out.cpu()[~touched] == 123.0).all())
    out2 = torch.full((P,), 5.0, device=DEV)
    ops.merge_rows(base.to(DEV), tv.to(DEV), alpha.to(DEV), torch.tensor([-1, V, 3], dtype=torch.int32, device=DEV), V, d, off, out2)
    t3 = torch.zeros(P, dtype=torch.bool); t3[off + 3 * d: off + 4 * d] = True
    assert torch.equal(out2.cpu()[t3], whole.cpu()[t3]) and bool((out2.cpu()[~t3] == 5.0).all())
    with pytest.raises(ValueError):
        ops.merge_rows(base.to(DEV), tv.to(DEV), alpha.to(DEV)[:3].contiguous(), idx.to(DEV), V, d, off, out)
    with pytest.raises(ValueError):
        ops.merge_rows(base.to(DEV), tv.to(DEV), alpha.to(DEV), idx.to(DEV), V, d, P - 64, out)


def test_pack_tokens_stores_clamped_indices_and_flags_the_original(ops):
    """the packed index arrays always address a row of their table (what the training graph's gathers / scatter-adds and the row-sparse merge
    read without clamping); the error word still reports the out-of-range original"""
    ids = torch.tensor([[0, 7, 60000, 2, 1], [0, -3, 5, 2, 1]], dtype=torch.int64)
    mask = torch.tensor([[1, 1, 1, 1, 0], [1, 1, 1, 1, 0]], dtype=torch.int64)
    tt = torch.tensor([[0, 1, 9, 2, 3], [0, 1, 2, 2, 3]], dtype=torch.int64)
    ip = torch.tensor([[0, 1, 1, 99, 0], [0, 1, 1, 1, 0]], dtype=torch.int64)
    cu = torch.tensor([0, 4, 8], dtype=torch.int32)
    err = torch.zeros(1, dtype=torch.int32, device=DEV)
    tw, tp, ttp, tip = ops.pack_tokens(ids.to(DEV), mask.to(DEV), cu.to(DEV), 8, 1, tt.to(DEV), ip.to(DEV), None, err, vocab=50265, n_type=4, n_ip=51)
    assert tw.cpu().tolist() == [0, 7, 50264, 2, 0, 0, 5, 2]
    assert ttp.cpu().tolist() == [0, 1, 3, 2, 0, 1, 2, 2] and tip.cpu().tolist() == [0, 1, 1, 50, 0, 1, 1, 1]
    bits = int(err.item())
    assert bits & 1 and bits & 4 and bits & 8  # MR_IN_BAD_ID | MR_IN_BAD_TOKEN_TYPE | MR_IN_BAD_ITEM_POS
