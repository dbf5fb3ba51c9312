"""float64 kernels (csrc/f64ops.hip, v_mfma_f64_16x16x4_f64) against torch's float64 arithmetic on the same inputs:
Gram SYRK over bf16 / fp32 activations, GEMM in every transpose combination with ragged sizes, G' = aG + (1-a)diag(G),
blocked Cholesky and the SPD right-hand solve RegMean uses in place of torch.inverse (vilt_module.py:432-434).
Tolerances are fp64 rounding: 1e-12 relative on sums of ~1e3 products, 1e-9 on the solve of a condition-1e4 system."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(pkg):
    return importlib.import_module("vl_merging_amd.ops")


@pytest.mark.parametrize("M,D,dtype", [(1000, 192, torch.bfloat16), (333, 768, torch.float32), (70, 100, torch.bfloat16),
                                       (5000, 3072, torch.bfloat16)])
def test_gram_f64_matches_torch(ops, M, D, dtype):
    g = torch.Generator(device="cuda"); g.manual_seed(M + D)
    x = torch.randn(M, D, device="cuda", generator=g).to(dtype)
    acc = torch.zeros(D, D, device="cuda", dtype=torch.float64)
    ops.gram_accumulate(x, acc)
    ops.gram_accumulate(x, acc)
    xd = x.double()
    want = 2 * (xd.t() @ xd)
    assert torch.allclose(acc, want, rtol=1e-12, atol=1e-9), float((acc - want).abs().max())
    assert torch.allclose(acc, acc.t(), rtol=1e-13, atol=1e-10)  # mirrored tiles (the row slices meet through atomics in any order)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("a32", [False, True])
def test_gemm_f64_matches_torch(ops, ta, tb, a32):
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    M, N, K = 150, 97, 333
    a = torch.randn((K, M) if ta else (M, K), device="cuda", generator=g, dtype=torch.float32 if a32 else torch.float64)
    b = torch.randn((N, K) if tb else (K, N), device="cuda", generator=g, dtype=torch.float64)
    c0 = torch.randn(M, N, device="cuda", generator=g, dtype=torch.float64)
    c = c0.clone()
    ops.gemm_f64(a, b, c, ta=ta, tb=tb, alpha=-0.5, beta=2.0)
    want = -0.5 * ((a.double().t() if ta else a.double()) @ (b.t() if tb else b)) + 2.0 * c0
    assert torch.allclose(c, want, rtol=1e-12, atol=1e-11), float((c - want).abs().max())
    # views with a leading dimension (the block updates of the factorisation)
    big = torch.zeros(M + 5, N + 9, device="cuda", dtype=torch.float64)
    ops.gemm_f64(a, b, big[2:2 + M, 3:3 + N], ta=ta, tb=tb)
    assert torch.allclose(big[2:2 + M, 3:3 + N], (a.double().t() if ta else a.double()) @ (b.t() if tb else b), rtol=1e-12, atol=1e-11)
    assert float(big[:2].abs().max()) == 0 and float(big[:, :3].abs().max()) == 0


def test_scale_gram(ops):
    g = torch.randn(130, 130, device="cuda", dtype=torch.float64)
    g = g @ g.t()
    out = torch.empty_like(g)
    ops.scale_gram(g, out, 0.9)
    want = 0.9 * g + (1 - 0.9) * torch.diag_embed(torch.diag(g))
    assert torch.equal(out, want)  # same two products and one add per element as the reference's expression
    ops.scale_gram(g, out, 0.9, accumulate=True)
    assert torch.equal(out, want + want)


# (264, 520, 600: the 256-column blocks of the drivers end in a ragged block / a ragged 64-column step; 17 / 50 rows: fewer than one
# workgroup of the triangular block solve)
@pytest.mark.parametrize("n,rows", [(64, 10), (200, 333), (264, 17), (520, 50), (600, 129), (768, 3072), (3072, 768)])
def test_cholesky_solve_matches_inverse(ops, n, rows):
    g = torch.Generator(device="cuda"); g.manual_seed(n)
    x = torch.randn(n + 64, n, device="cuda", generator=g, dtype=torch.float64)
    s = x.t() @ x                       # SPD, condition ~ 1e3..1e4 (the Gram matrices RegMean sums)
    num = torch.randn(rows, n, device="cuda", generator=g, dtype=torch.float64)
    chol = ops.cholesky_(s.clone())
    low = torch.tril(chol)
    assert torch.allclose(low @ low.t(), s, rtol=1e-12, atol=1e-9 * float(s.abs().max()))
    got = ops.solve_spd_right_(num.clone(), chol)
    want = num @ torch.inverse(s)       # what the reference computes
    scale = float(want.abs().max())
    assert float((got - want).abs().max()) <= 1e-9 * scale
    assert float((got @ s - num).abs().max()) <= 1e-9 * float(num.abs().max()) * (1 + n / 100)


def test_cholesky_rejects_indefinite(ops, pkg):
    L = importlib.import_module("vl_merging_amd._lib")
    s = torch.eye(100, device="cuda", dtype=torch.float64)
    s[70, 70] = -1.0
    with pytest.raises(L.VlmError):
        ops.cholesky_(s)


def test_batched_cholesky_and_solve_are_the_single_calls_bit_for_bit(ops):
    """vlm_cholesky_f64_batched / vlm_solve_spd_right_f64_batched (RegMean's solves of one shape in lock step, vilt_module.py:432-434):
    per matrix the same kernels in the same order as the single-matrix calls -- identical bits; a non-SPD member reports its pivot
    and leaves the others alone."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    n, rows, count = 200, 70, 5
    mats, rhs = [], []
    for i in range(count):
        x = torch.randn(n + 40, n, device="cuda", dtype=torch.float64, generator=gen)
        mats.append((x.t() @ x).contiguous())
        rhs.append(torch.randn(rows, n, device="cuda", dtype=torch.float64, generator=gen))
    mats[3] = mats[3] - 1e3 * torch.eye(n, device="cuda", dtype=torch.float64)  # indefinite
    single_c, single_x = [], []
    for i in range(count):
        st = torch.zeros(1, device="cuda", dtype=torch.int32)
        c = ops.cholesky_(mats[i].clone(), status=st)
        single_c.append((c, int(st.item())))
        single_x.append(ops.solve_spd_right_(rhs[i].clone(), c))
    status = torch.zeros(count, device="cuda", dtype=torch.int32)
    bc = ops.cholesky_batched_([m.clone() for m in mats], status)
    bx = ops.solve_spd_right_batched_([r.clone() for r in rhs], bc)
    torch.cuda.synchronize()
    verdict = status.tolist()
    for i in range(count):
        assert verdict[i] == single_c[i][1]
        if i == 3:
            assert verdict[i] > 0
            continue
        assert verdict[i] == 0
        assert torch.equal(torch.tril(bc[i]), torch.tril(single_c[i][0])), i
        assert torch.equal(bx[i], single_x[i]), i
        want = rhs[i] @ torch.linalg.inv(mats[i])
        assert float((bx[i] - want).abs().max()) <= 1e-8 * float(want.abs().max())


@pytest.mark.parametrize("a32", [False, True])
def test_batched_gemm_is_the_single_call_bit_for_bit(ops, a32):
    """vlm_gemm_f64_batched (RegMean's W_m G'_m products of one shape in one launch, vilt_module.py:421-423): per product the bits of
    vlm_gemm_f64, with and without the accumulate (second model's term); 70 products cross the 64-per-launch table."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    M, K, N, count = 100, 136, 72, 70
    A = [torch.randn(M, K, device="cuda", dtype=torch.float32 if a32 else torch.float64, generator=gen) for _ in range(count)]
    B = [torch.randn(K, N, device="cuda", dtype=torch.float64, generator=gen) for _ in range(count)]
    C0 = [torch.randn(M, N, device="cuda", dtype=torch.float64, generator=gen) for _ in range(count)]
    for beta in (0.0, 1.0):
        single = [ops.gemm_f64(a, b, c.clone(), beta=beta) for a, b, c in zip(A, B, C0)]
        got = ops.gemm_f64_batched(A, B, [c.clone() for c in C0], beta=beta)
        for i in range(count):
            assert torch.equal(got[i], single[i]), (beta, i)
        want = A[7].double() @ B[7] + beta * C0[7]
        assert float((got[7] - want).abs().max()) <= 1e-12 * float(want.abs().max())


def test_gram_slices_fill_whole_rounds(ops):
    """vlm_gram_f64 at the two capture widths with a row count that gives every slice count a chance: the accumulated G equals the
    fp64 product of the bf16 rows whatever the slice count chosen for the launch."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(6)
    for D, rows in ((768, 3001), (3072, 1500)):
        x = torch.randn(rows, D, device="cuda", generator=gen).to(torch.bfloat16)
        g = torch.zeros(D, D, device="cuda", dtype=torch.float64)
        ops.gram_accumulate(x, g)
        want = x.double().t() @ x.double()
        assert float((g - want).abs().max()) <= 1e-11 * float(want.abs().max())
        assert torch.equal(g, g.t())
