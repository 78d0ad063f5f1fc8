"""The plain-C part of the oracle (oracle/oracle_c.c) pinned on the CPU: against the reference's own outputs in the golden fixtures (merge: g2,
top-k: g1), against the torch restatement (oracle/ref_cpu.py) and against exact arithmetic.  The GPU tests use these functions as the
bit-level checker of the merge, scoring-GEMM and selection kernels, so they are checked here before anything is trusted on the device."""
from fractions import Fraction

import numpy as np
import torch

from oracle import c_oracle as CO
from oracle import ref_cpu as O
from tests.conftest import load_golden


def _alpha(case, key, n):
    w = case["weights"]
    return O.effective_alpha(torch.tensor(w["global_weights"][key]), torch.tensor(w["global_biases"][key]), torch.tensor(w["per_weights"][key])[:n],
                             disable_softmax=not case["use_softmax"])


def test_c_merge_equals_the_reference_modules_output():
    """merge_nway_ref against TaskVectorMergingModule{TaskWise,LayerWise}.get_state_dict() of the reference (fixture g2), bit for bit."""
    g2 = load_golden("g2_merger.pt")
    pre, fts = O.align_state_dicts(g2["pretrain"], g2["finetunes"])
    base, shape_dict = O.flatten_model(pre)
    tv = O.get_task_vectors(base, [O.flatten_model(ft)[0] for ft in fts])
    n = tv.shape[0]
    seen = set()
    for case in g2["cases"]:
        seen.add(case["learn_type"])
        if case["learn_type"] == "TASK_WISE":
            got = CO.merge_nway(base, tv, _alpha(case, "all", n))
        else:  # one segment per tensor (layer_wise.py:64-83 walks (name, start, end) chunks), alpha row = the tensor's group
            groups = O.group_parameters_by_layer(shape_dict)
            chunks = sorted((s, e, key) for key, cs in groups.items() for _, s, e in cs)
            assert chunks[0][0] == 0 and all(a[1] == b[0] for a, b in zip(chunks, chunks[1:])) and chunks[-1][1] == base.numel()
            seg_off = torch.tensor([c[0] for c in chunks] + [base.numel()], dtype=torch.int64)
            alpha = torch.stack([_alpha(case, key, n) for _, _, key in chunks])
            got = CO.merge_nway(base, tv, alpha, seg_off)
        assert torch.equal(got, case["merged_flat"]), case["learn_type"]
    assert seen == {"TASK_WISE", "LAYER_WISE"}


def test_c_merge_rounds_every_product_and_sums_in_task_order():
    """No fused multiply-add, sequential sum from 0: checked against exact rational arithmetic with one rounding per operation."""
    g = torch.Generator().manual_seed(3)
    N, P = 5, 64
    base, tv, alpha = torch.randn(P, generator=g), torch.randn(N, P, generator=g) * 1e-3, torch.rand(N, generator=g)
    got = CO.merge_nway(base, tv, alpha)
    for p in range(P):
        acc = np.float32(0.0)
        for i in range(N):
            prod = np.float32(alpha[i].item()) * np.float32(tv[i, p].item())   # numpy scalar ops round once, like the C expression
            acc = np.float32(acc + prod)
        want = np.float32(np.float32(base[p].item()) + acc)
        assert got[p].item() == float(want), p
        exact = Fraction(base[p].item()) + sum(Fraction(alpha[i].item()) * Fraction(tv[i, p].item()) for i in range(N))
        assert abs(Fraction(float(got[p])) - exact) < Fraction(1, 2 ** 20) * max(1, abs(exact))
    assert torch.equal(got, O.merge_task_wise(base, tv, alpha))


def test_c_topk_equals_the_reference_evaluators_topk():
    """topk_rows_ref against torch.topk as the reference's Evaluator called it (fixture g1): the same values everywhere, the same indices
    wherever a row's values are distinct, ascending index among equal values."""
    for case in load_golden("g1_evaluator.pt"):
        k = case["ref_topk_idx"].shape[1]
        val, idx = CO.topk_rows(case["scores"], k)
        assert torch.equal(val, case["ref_topk_val"])
        oval, oidx = O.topk_canonical(case["scores"], k)
        assert torch.equal(val, oval) and torch.equal(idx, oidx), "the C and the torch restatement agree on the canonical order"
        for r in range(val.shape[0]):
            if len(torch.unique(val[r])) == k:
                assert torch.equal(idx[r], case["ref_topk_idx"][r])


def test_c_topk_ties_nan_and_short_rows():
    s = torch.tensor([[0.1, float("nan"), 0.5, 0.5, float("nan"), -1.0], [2.0, 2.0, 2.0, 2.0, 2.0, 2.0], [0.0, -0.0, 1.0, -1.0, 0.0, -0.0]])
    val, idx = CO.topk_rows(s, 4)
    assert idx.tolist() == [[1, 4, 2, 3], [0, 1, 2, 3], [2, 0, 1, 4]], "NaN first, then score descending, index ascending among equals (+0 == -0)"
    assert torch.isnan(val[0, :2]).all() and val[0, 2:].tolist() == [0.5, 0.5]
    val, idx = CO.topk_rows(s[:, :3], 3)   # k == ncols: a full sort of the row
    assert idx.tolist() == [[1, 2, 0], [0, 1, 2], [2, 0, 1]]


def test_c_gemm_is_one_fma_chain_in_ascending_k():
    # integer-valued operands: every partial sum is exact, so the result is the exact product whatever the rounding mode of the chain
    g = torch.Generator().manual_seed(5)
    A = torch.randint(-8, 9, (7, 48), generator=g).float()
    W = torch.randint(-8, 9, (5, 48), generator=g).float()
    b = torch.randint(-4, 5, (5,), generator=g).float()
    assert torch.equal(CO.gemm_nt(A, W, b), (A.double() @ W.double().T + b.double()).float())
    # real-valued operands: each step is round(a * w + acc) with ONE rounding -- reproduced with exact rationals
    A, W = torch.randn(3, 40, generator=g), torch.randn(4, 40, generator=g)
    got = CO.gemm_nt(A, W)
    for m in range(3):
        for n in range(4):
            acc = Fraction(0)
            for k in range(40):
                exact = Fraction(A[m, k].item()) * Fraction(W[n, k].item()) + acc
                acc = Fraction(float(np.float32(_round_to_f32(exact))))
            assert got[m, n].item() == float(acc), (m, n)
    assert float((got.double() - A.double() @ W.double().T).abs().max()) < 1e-5


def _round_to_f32(q: Fraction) -> float:
    """nearest float32 (ties to even) of an exact rational, without passing through a double rounding"""
    if q == 0:
        return 0.0
    sign = -1 if q < 0 else 1
    a = abs(q)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    e = max(e, -126)                                   # subnormals share the exponent of the smallest normal
    scaled = a / Fraction(2) ** (e - 23)               # the integer part is the 24-bit significand
    n = scaled.numerator // scaled.denominator
    rem = scaled - n
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and n % 2 == 1):
        n += 1
    return sign * float(n) * 2.0 ** (e - 23)
