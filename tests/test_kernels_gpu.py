"""HIP kernels (through the C ABI) vs plain PyTorch fp32 references of the same op on the same inputs.

Tolerances: inputs are bf16-exact (the references see the same bf16-rounded values in fp32), accumulation is
fp32, so the only differences are summation order and the final rounding of a bf16 output:
  f32 outputs : atol 2e-3 * scale, rtol 1e-3      bf16 outputs: rtol 2^-7 (one bf16 ulp) on top.
"""
import importlib
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(pkg):
    return importlib.import_module("vl_merging_amd.ops")


@pytest.fixture(scope="module")
def L(pkg):
    return importlib.import_module("vl_merging_amd._lib")


def bf(x):
    return x.to(torch.bfloat16)


def assert_close(got, ref, rtol, atol, what=""):
    got = got.float()
    ref = ref.float()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    assert not bad.any(), "%s: %d/%d bad, max err %.4g (ref max %.4g)" % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(ref.abs().max()))


def make_ab(M, N, K, ta, tb, gen):
    # integer-valued asymmetric data first (exact in bf16 and in fp32 accumulation) then random
    a = torch.randn(M, K, device="cuda", generator=gen)
    b = torch.randn(N, K, device="cuda", generator=gen)
    A = bf(a.t().contiguous() if ta else a)
    B = bf(b.t().contiguous() if tb else b)
    a32 = (A.float().t() if ta else A.float())
    b32 = (B.float().t() if tb else B.float())
    return A, B, a32 @ b32.t()


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (880, 768, 768), (577 * 3, 2304, 768), (333, 3072, 768),
                                   (130, 72, 128), (1, 8, 64), (2, 2, 768)])
def test_gemm_layouts(ops, ta, tb, M, N, K):
    gen = torch.Generator(device="cuda"); gen.manual_seed(M * 7 + N * 3 + K)
    A, B, ref = make_ab(M, N, K, ta, tb, gen)
    if (ta and (M % 8)) or (tb and (N % 8)):
        pytest.skip("leading dimension of a K-strided operand must be a multiple of 8")
    out = torch.empty(M, N, device="cuda", dtype=torch.float32)
    ops.gemm(A, B, out, ta, tb)
    assert_close(out, ref, 1e-3, 2e-3 * math.sqrt(K), "gemm f32")
    outb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, outb, ta, tb)
    assert_close(outb, ref, 1e-2, 2e-3 * math.sqrt(K), "gemm bf16")


def test_gemm_exact_integers_asymmetric(ops):
    """A = I-like / small integers with an ASYMMETRIC B: catches row/col swaps and k-order permutations."""
    M, N, K = 256, 256, 128
    a = torch.zeros(M, K, device="cuda")
    a[torch.arange(M), torch.arange(M) % K] = 1.0
    a[:, 0] += (torch.arange(M, device="cuda") % 3).float()
    b = (torch.arange(N, device="cuda").view(N, 1) * 2 + torch.arange(K, device="cuda").view(1, K) * 5) % 17
    b = b.float() - 8
    for ta in (False, True):
        for tb in (False, True):
            A = bf(a.t().contiguous() if ta else a)
            B = bf(b.t().contiguous() if tb else b)
            out = torch.empty(M, N, device="cuda", dtype=torch.float32)
            ops.gemm(A, B, out, ta, tb)
            assert torch.equal(out, a @ b.t()), (ta, tb)


def test_gemm_kstrided_ragged_k(ops):
    """wgrad shape: reduction over tokens (K = 1357, not a multiple of 64), both operands K-strided."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    K, M, N = 1357, 768, 384
    dy = bf(torch.randn(K, M, device="cuda", generator=gen))
    x = bf(torch.randn(K, N, device="cuda", generator=gen))
    ref = dy.float().t() @ x.float()
    out = torch.full((M, N), 1.0, device="cuda")
    ops.gemm(dy, x, out, True, True, accumulate=True)
    assert_close(out, ref + 1.0, 1e-3, 2e-3 * math.sqrt(K), "wgrad accumulate")
    # a row-range view of a taller matrix: rows beyond the range must not leak in
    big = bf(torch.randn(K + 300, M, device="cuda", generator=gen))
    bigx = bf(torch.randn(K + 300, N, device="cuda", generator=gen))
    out2 = torch.empty(M, N, device="cuda")
    ops.gemm(big[100:100 + K], bigx[100:100 + K], out2, True, True)
    assert_close(out2, big[100:100 + K].float().t() @ bigx[100:100 + K].float(), 1e-3, 2e-3 * math.sqrt(K), "range")


@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("M,N,K", [(13574, 3072, 768), (13574, 2304, 768), (54296, 768, 3072), (13574 * 2, 1536, 64)])
def test_gemm_training_shapes_with_epilogues(ops, L, tb, M, N, K):
    """The 128x128 kernel at the training shapes of the bench (ragged M: 13574 = 106*128 + 6; the 4B-sample pass's
    M = 54296; a K = 64 head-sized reduction) with the fused epilogues the block function uses."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(M + N + K + tb)
    A, B, ref = make_ab(M, N, K, False, tb, gen)
    ref = ref * 0.05
    bias = torch.randn(N, device="cuda", generator=gen)
    res = torch.randn(M, N, device="cuda", generator=gen)
    gamma = torch.randn(N, device="cuda", generator=gen) * 0.1
    out = torch.empty(M, N, device="cuda")
    aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, out, False, tb, bias=bias, col_scale=gamma, residual=res, aux=aux, alpha=0.05)
    assert_close(aux, ref + bias, 1e-2, 1e-2, "big aux")
    assert_close(out, res + gamma[None] * (ref + bias), 1e-3, 5e-3, "big epilogue")
    h = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, h, False, tb, bias=bias, act=L.ACT_GELU, alpha=0.05)
    assert_close(h, torch.nn.functional.gelu(ref + bias), 1e-2, 1e-2, "big gelu")


def test_gemm_wgrad_splitk(ops):
    """wgrad at the training shape: reduction over 13 574 tokens into 6 x 24 output tiles -> split-K with fp32 atomic
    accumulation (order-dependent rounding: tolerance as for any fp32 sum of ~13.5k terms)."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(17)
    K, M, N = 13574, 768, 3072
    dy = bf(torch.randn(K, M, device="cuda", generator=gen))
    x = bf(torch.randn(K, N, device="cuda", generator=gen))
    ref = dy.float().t() @ x.float()
    out = torch.full((M, N), 2.0, device="cuda")
    ops.gemm(dy, x, out, True, True, accumulate=True)
    assert_close(out, ref + 2.0, 1e-3, 2e-3 * math.sqrt(K), "split-K wgrad")
    out2 = torch.zeros(M, 264, device="cuda")  # ragged N tile, narrow output
    ops.gemm(dy, x[:, :264], out2, True, True, accumulate=True, alpha=0.5)
    assert_close(out2, 0.5 * ref[:, :264], 1e-3, 2e-3 * math.sqrt(K), "split-K ragged")


def test_gemm_dgrad_splitk_long_reduction(ops):
    """The MLM decoder's dgrad shape: [880 x 30 528] . [30 528 x 768] -- 42 output tiles, a reduction of 30 528: with
    accumulate=True into fp32 the library cuts K over one round of workgroups (ta = 0, tb = 1 split-K)."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(23)
    M, N, K = 880, 768, 30528
    a = bf(torch.randn(M, K, device="cuda", generator=gen) * 0.1)
    w = bf(torch.randn(K, N, device="cuda", generator=gen) * 0.1)
    ref = a.float() @ w.float()
    out = torch.full((M, N), -1.0, device="cuda")
    ops.gemm(a, w, out, False, True, accumulate=True)
    assert_close(out, ref - 1.0, 1e-3, 2e-5 * math.sqrt(K), "split-K dgrad")
    out_b = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)   # the unsplit path of the same product
    ops.gemm(a, w, out_b, False, True)
    assert_close(out_b.float(), ref, 1e-2, 1e-2, "unsplit dgrad")


def test_gemm_epilogues(ops, L):
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    M, N, K = 617 * 2, 768, 768
    A, B, acc = make_ab(M, N, K, False, False, gen)
    acc = acc * 0.05
    bias = torch.randn(N, device="cuda", generator=gen)
    gamma = torch.randn(N, device="cuda", generator=gen) * 0.1
    rs = (torch.rand(M, device="cuda", generator=gen) > 0.2).float() / 0.8
    res = torch.randn(M, N, device="cuda", generator=gen)
    # proj / fc2 epilogue: residual + rs*gamma*(acc+bias), aux = bf16(acc+bias)
    out = torch.empty(M, N, device="cuda")
    aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, out, bias=bias, col_scale=gamma, row_scale=rs, residual=res, aux=aux, alpha=0.05)
    y = acc + bias
    assert_close(aux, y, 1e-2, 1e-2, "aux")
    assert_close(out, res + rs[:, None] * gamma[None] * y, 1e-3, 5e-3, "layerscale epilogue")
    # fc1 epilogue: gelu(acc+bias) in bf16, preact saved
    h = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    a = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, a, bias=bias, act=L.ACT_GELU, aux=h, alpha=0.05)
    assert_close(h, y, 1e-2, 1e-2, "preact")
    assert_close(a, torch.nn.functional.gelu(y), 1e-2, 1e-2, "gelu")
    # dgrad with GELU backward: (acc) * gelu'(h)
    d = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, d, act=L.ACT_GELU_BWD, aux=h, alpha=0.05)
    hh = h.float().requires_grad_(True)
    torch.nn.functional.gelu(hh).backward(acc)
    assert_close(d, hh.grad, 1e-2, 1e-2, "gelu bwd")
    # fused bias gradient: col_sum += column sums of what the epilogue wrote (M = 1234 has interior AND edge tiles)
    cs = torch.full((N,), 3.0, device="cuda")
    ops.gemm(A, B, d, act=L.ACT_GELU_BWD, aux=h, alpha=0.05, col_sum=cs)
    assert_close(cs - 3.0, hh.grad.sum(0), 2e-3, 2e-2, "col_sum (gelu bwd, bf16 out)")
    cs32 = torch.zeros(N, device="cuda")
    o32 = torch.empty(M, N, device="cuda")
    ops.gemm(A, B, o32, bias=bias, alpha=0.05, col_sum=cs32)
    assert_close(cs32, y.sum(0), 2e-3, 2e-2, "col_sum (f32 out)")
    # ragged N (scalar tail path)
    cs_r = torch.zeros(N, device="cuda")
    o_r = torch.empty(M, 50, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B[:50], o_r, alpha=0.05, col_sum=cs_r)
    assert_close(cs_r[:50], acc[:, :50].sum(0), 2e-3, 2e-2, "col_sum (ragged N)")


def test_gemm_large_m_epilogues(ops, L):
    """Training-sized M (thousands of tiles, grouped raster, ragged last tile) with every epilogue of the transformer
    block, against fp32 matmul of the bf16 operands."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(21)
    M, N, K = 256 * 130 + 77, 2304, 768
    A = bf(torch.randn(M, K, device="cuda", generator=gen))
    B = bf(torch.randn(N, K, device="cuda", generator=gen))
    ref = (A.float() @ B.float().t()) * 0.05
    bias = torch.randn(N, device="cuda", generator=gen)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, out, bias=bias, alpha=0.05)
    assert_close(out, ref + bias, 1e-2, 2e-2, "wide bf16+bias")
    gamma = torch.randn(N, device="cuda", generator=gen) * 0.1
    res = torch.randn(M, N, device="cuda", generator=gen)
    want = res + gamma[None] * (ref + bias)
    ops.gemm(A, B, res, bias=bias, col_scale=gamma, residual=res, alpha=0.05)  # in-place residual stream
    assert_close(res, want, 1e-3, 5e-3, "wide f32 residual")
    h = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    a = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, a, bias=bias, act=L.ACT_GELU, aux=h, alpha=0.05)
    assert_close(h, ref + bias, 1e-2, 2e-2, "wide preact")
    assert_close(a, torch.nn.functional.gelu(ref + bias), 1e-2, 2e-2, "wide gelu")
    # K = 64 (two 32-deep K steps) and 128 (four): pipeline prologue / tail
    for k in (64, 128):
        o2 = torch.empty(M, N, device="cuda")
        ops.gemm(A[:, :k], B[:, :k], o2)
        assert_close(o2, A[:, :k].float() @ B[:, :k].float().t(), 1e-3, 2e-3 * math.sqrt(k), "wide K=%d" % k)


def test_gemm_vocab_ragged_n(ops):
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    M, N, K, NP = 80, 30522, 768, 30528
    x = bf(torch.randn(M, K, device="cuda", generator=gen))
    w = bf(torch.randn(N, K, device="cuda", generator=gen) * 0.05)
    bias = torch.randn(N, device="cuda", generator=gen)
    buf = torch.zeros(M, NP, device="cuda", dtype=torch.bfloat16)
    ops.gemm(x, w, buf[:, :N], bias=bias)
    assert_close(buf[:, :N], x.float() @ w.float().t() + bias, 1e-2, 2e-2, "vocab logits")
    assert float(buf[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("M,D", [(1, 192), (37, 192), (880, 768), (12694, 768), (5, 1024)])
@pytest.mark.parametrize("out_f32", [False, True])
def test_layernorm(ops, M, D, out_f32):
    gen = torch.Generator(device="cuda"); gen.manual_seed(M + D)
    x = torch.randn(M, D, device="cuda", generator=gen) * 3 + 0.5
    g = 1 + 0.1 * torch.randn(D, device="cuda", generator=gen)
    b = 0.1 * torch.randn(D, device="cuda", generator=gen)
    y = torch.empty(M, D, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
    stats = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(x, g, b, 1e-6, y, stats)
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    assert_close(y, ref, 1e-2 if not out_f32 else 1e-5, 1e-2 if not out_f32 else 1e-5, "ln fwd")
    dy = torch.randn(M, D, device="cuda", generator=gen)
    dres = torch.randn(M, D, device="cuda", generator=gen)
    dyq = dy if out_f32 else bf(dy)
    ref.backward(dyq.float())
    dx = torch.empty(M, D, device="cuda")
    dg = torch.zeros(D, device="cuda")
    db = torch.zeros(D, device="cuda")
    ops.layernorm_bwd(dyq, x, stats, g, dx, dres=dres, dgamma=dg, dbeta=db)
    assert_close(dx, xr.grad + dres, 1e-4, 1e-4, "ln dx")
    assert_close(dg, gr.grad, 1e-4, 1e-3 * math.sqrt(M), "ln dgamma")
    assert_close(db, br.grad, 1e-4, 1e-3 * math.sqrt(M), "ln dbeta")


def test_layerscale_bwd_and_colsum(ops):
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    M, D = 1234, 768
    dx = torch.randn(M, D, device="cuda", generator=gen)
    y = bf(torch.randn(M, D, device="cuda", generator=gen))
    gamma = torch.randn(D, device="cuda", generator=gen) * 0.1
    rs = (torch.rand(M, device="cuda", generator=gen) > 0.3).float() / 0.7
    dy = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    dg = torch.zeros(D, device="cuda"); dbias = torch.zeros(D, device="cuda")
    ops.layerscale_bwd(dx, y, gamma, rs, dy, dg, dbias)
    ref_dy = rs[:, None] * gamma[None] * dx
    assert_close(dy, ref_dy, 1e-2, 1e-4, "dy")
    assert_close(dg, (rs[:, None] * dx * y.float()).sum(0), 1e-4, 1e-2, "dgamma")
    assert_close(dbias, dy.float().sum(0), 1e-4, 1e-3, "dbias")
    a = bf(torch.randn(777, 2304, device="cuda", generator=gen))
    out = torch.ones(2304, device="cuda")
    ops.colsum(a, out)
    assert_close(out, a.float().sum(0) + 1, 1e-4, 1e-3, "colsum")
    out2 = torch.zeros(768, device="cuda")
    ops.colsum(a[:, 1536:], out2)
    assert_close(out2, a[:, 1536:].float().sum(0), 1e-4, 1e-3, "colsum slice")


@pytest.mark.parametrize("M,D", [(1, 192), (37, 192), (880, 768), (13574, 768), (9, 1024)])
@pytest.mark.parametrize("fold,with_rs", [(False, False), (True, True)])
def test_layernorm_bwd_scale_is_the_two_calls(ops, M, D, fold, with_rs):
    """vlm_layernorm_bwd_scale = vlm_layernorm_bwd, then vlm_layerscale_bwd on the row it produced: every output bit for bit
    (dx, the branch's bf16 gradient; the four column sums to the order of their partial sums when folded from the same grid)."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(M * 7 + D)
    x = torch.randn(M, D, device="cuda", generator=gen) * 2 + 0.3
    g = 1 + 0.1 * torch.randn(D, device="cuda", generator=gen)
    stats = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(x, g, torch.zeros(D, device="cuda"), 1e-6, torch.empty(M, D, device="cuda", dtype=torch.bfloat16), stats)
    dy = bf(torch.randn(M, D, device="cuda", generator=gen))
    dres = torch.randn(M, D, device="cuda", generator=gen)
    y = bf(torch.randn(M, D, device="cuda", generator=gen))
    sg = torch.randn(D, device="cuda", generator=gen) * 0.1
    rs = ((torch.rand(M, device="cuda", generator=gen) > 0.3).float() / 0.7) if with_rs else None

    def run(fused):
        dx = torch.empty(M, D, device="cuda"); sdy = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        sums = [torch.full((D,), 0.5, device="cuda") for _ in range(4)]
        fb = ops.FoldBatch(x.device, D) if fold else None
        if fused:
            ops.layernorm_bwd_scale(dy, x, stats, g, dx, dres, sums[0], sums[1], y=y, sgamma=sg, row_scale=rs, sdy=sdy,
                                    dsgamma=sums[2], dsbias=sums[3], fold=fb)
        else:
            ops.layernorm_bwd(dy, x, stats, g, dx, dres=dres, dgamma=sums[0], dbeta=sums[1], fold=fb)
            ops.layerscale_bwd(dx, y, sg, rs, sdy, sums[2], sums[3], fold=fb)
        if fb is not None:
            fb.flush()
        torch.cuda.synchronize()
        return dx, sdy, sums

    dx0, sdy0, s0 = run(False)
    dx1, sdy1, s1 = run(True)
    assert torch.equal(dx0, dx1) and torch.equal(sdy0, sdy1)
    for a, b, name in zip(s0, s1, ("dgamma", "dbeta", "dsgamma", "dsbias")):
        assert_close(b, a, 1e-5, 1e-4 * math.sqrt(M), name)  # the fused launch has half the workgroups: other partial sums


def test_adamw_matches_hf4_rule(ops):
    gen = torch.Generator(device="cuda"); gen.manual_seed(2)
    n = 4096 * 3 + 4
    p = torch.randn(n, device="cuda", generator=gen); g = torch.randn(n, device="cuda", generator=gen)
    m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    pb = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    pr, mr, vr = p.double().clone(), m.double().clone(), v.double().clone()
    lr, b1, b2, eps, wd = 2e-4, 0.9, 0.98, 1e-8, 0.01
    for step in (1, 2, 3):
        gg = g * (1.0 + 0.1 * step)
        ops.adamw_step(p, gg.clone(), m, v, pb, lr, b1, b2, eps, wd, step, grad_scale=0.5, zero_grad=False)
        gd = gg.double() * 0.5
        mr = b1 * mr + (1 - b1) * gd
        vr = b2 * vr + (1 - b2) * gd * gd
        ss = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        pr = pr - ss * mr / (vr.sqrt() + eps)
        pr = pr - lr * wd * pr
    assert_close(p, pr.float(), 1e-5, 1e-6, "adamw p")
    assert torch.equal(pb, p.to(torch.bfloat16))


def test_im2col_matches_conv(ops):
    gen = torch.Generator(device="cuda"); gen.manual_seed(4)
    B, H, P, D = 3, 64, 16, 64
    img = torch.rand(B, 3, H, H, device="cuda", generator=gen) * 2 - 1
    w = torch.randn(D, 3, P, P, device="cuda", generator=gen) * 0.05
    bias = torch.randn(D, device="cuda", generator=gen)
    np_ = (H // P) ** 2
    patches = torch.empty(B * (np_ + 1), 3 * P * P, device="cuda", dtype=torch.bfloat16)
    ops.patch_im2col(img, patches, P, 1)
    out = torch.empty(B * (np_ + 1), D, device="cuda")
    ops.gemm(patches, bf(w.view(D, -1)), out, bias=bias)
    ref = torch.nn.functional.conv2d(bf(img).float(), bf(w).float(), bias, stride=P).flatten(2).transpose(1, 2)
    got = out.view(B, np_ + 1, D)
    assert_close(got[:, 1:], ref, 1e-3, 2e-3, "patch embed")
    assert_close(got[:, 0], bias.expand(B, D), 0, 0, "lead row = bias")


def test_droppath_rows(ops):
    """vlm_droppath_rows == timm drop_path's per-sample mask/keep expanded to the segment-major rows."""
    B, n0, n1 = 5, 7, 13
    seq = ops.Seq(B, n0, n1)
    u = torch.tensor([0.05, 0.95, 0.5, 0.899, 0.9], device="cuda")
    out = torch.full((seq.rows,), -1.0, device="cuda")
    ops.droppath_rows(u, 0.9, seq, out)
    per = torch.where(u < 0.9, torch.tensor(1.0 / 0.9, device="cuda"), torch.tensor(0.0, device="cuda"))
    want = torch.cat([per.repeat_interleave(n0), per.repeat_interleave(n1)])
    assert torch.equal(out, want)
    # image-only layout (n0 = 0)
    seq2 = ops.Seq(B, 0, n1)
    out2 = torch.empty(seq2.rows, device="cuda")
    ops.droppath_rows(u, 0.9, seq2, out2)
    assert torch.equal(out2, per.repeat_interleave(n1))


def test_transpose_tiles(ops):
    """Batched bf16 transpose: several matrices (one ragged) in one launch, bit-exact."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(9)
    shapes = [(768, 2304), (192, 64), (100, 70), (64, 64)]
    srcs = [bf(torch.randn(r, c, device="cuda", generator=gen)) for r, c in shapes]
    dsts = [torch.full((c, r), 7.0, device="cuda", dtype=torch.bfloat16) for r, c in shapes]
    table, n = ops.transpose_table(list(zip(srcs, dsts)))
    assert n == 12 * 36 + 3 * 1 + 2 * 2 + 1
    ops.transpose_tiles(table, n)
    for s_, d_ in zip(srcs, dsts):
        assert torch.equal(d_, s_.t().contiguous())


def test_deferred_fold_batch(ops):
    """Row kernels with a deferred fold + ONE vlm_colreduce_batch == the same kernels folding on their own."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(13)
    M, D = 3000, 768
    xs = torch.randn(M, D, device="cuda", generator=gen)
    dy = bf(torch.randn(M, D, device="cuda", generator=gen))
    gamma = torch.randn(D, device="cuda", generator=gen)
    stats = torch.empty(M, 2, device="cuda")
    y = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    ops.layernorm_fwd(xs, gamma, torch.zeros(D, device="cuda"), 1e-6, y, stats)
    dxs = torch.randn(M, D, device="cuda", generator=gen)
    ybr = bf(torch.randn(M, D, device="cuda", generator=gen))
    outs = []
    for use_fold in (False, True):
        fold = ops.FoldBatch(xs.device, D) if use_fold else None
        dg, db = torch.full((D,), 2.0, device="cuda"), torch.zeros(D, device="cuda")
        dg2, dbias = torch.zeros(D, device="cuda"), torch.full((D,), -1.0, device="cuda")
        dx = torch.empty(M, D, device="cuda")
        dyo = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        ops.layernorm_bwd(dy, xs, stats, gamma, dx, dgamma=dg, dbeta=db, fold=fold)
        ops.layerscale_bwd(dxs, ybr, gamma, None, dyo, dg2, dbias, fold=fold)
        if use_fold:
            assert float(db.abs().max()) == 0.0  # nothing folded yet
            fold.flush()
        outs.append((dg, db, dg2, dbias, dx, dyo))
    for a, b in zip(*outs):
        assert_close(a, b, 1e-4, 1e-3 * float(b.float().abs().max()) + 1e-6, "deferred fold")


def test_gemm_col_sum_through_fold_workspace(ops, L):
    """col_sum with a FoldBatch: complete 128-row tiles park their column sums in the workspace, the ragged last tile adds
    directly, one vlm_colreduce_batch folds -- same result as the direct col_sum and as a torch column sum."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(31)
    M, N, K = 128 * 9 + 50, 256, 128
    A = bf(torch.randn(M, K, device="cuda", generator=gen))
    B = bf(torch.randn(N, K, device="cuda", generator=gen))
    h = bf(torch.randn(M, N, device="cuda", generator=gen))
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    direct = torch.full((N,), 1.5, device="cuda")
    ops.gemm(A, B, out, act=L.ACT_GELU_BWD, aux=h, alpha=0.1, col_sum=direct)
    fold = ops.FoldBatch(A.device, N)
    viafold = torch.full((N,), 1.5, device="cuda")
    ops.gemm(A, B, out, act=L.ACT_GELU_BWD, aux=h, alpha=0.1, col_sum=viafold, col_sum_fold=fold)
    assert len(fold.jobs) == 1 and fold.jobs[0][1] == 9
    part = viafold.clone()
    fold.flush()
    hh = h.float().requires_grad_(True)
    torch.nn.functional.gelu(hh).backward((A.float() @ B.float().t()) * 0.1)
    want = hh.grad.sum(0) + 1.5
    assert_close(direct, want, 2e-3, 2e-2, "direct col_sum")
    assert_close(viafold, want, 2e-3, 2e-2, "col_sum through the fold workspace")
    assert float((part - 1.5).abs().max()) < float((want - 1.5).abs().max())  # before the fold: only the ragged tile's 50 rows


@pytest.fixture
def big_tile(L):
    """Force the 256x256 kernel wherever it is legal for the duration of a test, then hand the choice back."""
    lib = L.get_lib()
    L.check(lib.vlm_gemm_set_big_tile_mode(2), "vlm_gemm_set_big_tile_mode")
    yield lib
    L.check(lib.vlm_gemm_set_big_tile_mode(-1), "vlm_gemm_set_big_tile_mode")


@pytest.mark.parametrize("M,N,K", [(256 * 5 + 77, 768, 768), (13574, 2304, 768), (256 * 3 + 130, 256, 128), (1000, 3072, 3072),
                                   (256 * 2 + 9, 384, 256), (700, 1152, 128)])  # N % 256 == 128: must not reach the 256x256 kernel
def test_gemm_big_tile_every_epilogue_variant(ops, L, big_tile, M, N, K):
    """The 256x256 kernel's looped epilogue (LDS transpose, buffer loads/stores, rows >= M falling off the descriptors)
    in every variant the launcher offers, against fp32 matmul of the bf16 operands AND against the 128x128 kernel."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(M + N + K)
    A = bf(torch.randn(M, K, device="cuda", generator=gen))
    B = bf(torch.randn(N, K, device="cuda", generator=gen))
    alpha = 1.0 / math.sqrt(K)
    ref = (A.float() @ B.float().t()) * alpha
    bias = torch.randn(N, device="cuda", generator=gen)
    gamma = torch.randn(N, device="cuda", generator=gen) * 0.5
    rs = (torch.rand(M, device="cuda", generator=gen) > 0.3).float() / 0.7
    res = torch.randn(M, N, device="cuda", generator=gen)
    hpre = bf(torch.randn(M, N, device="cuda", generator=gen))

    def run(mode):
        L.check(big_tile.vlm_gemm_set_big_tile_mode(mode), "mode")
        r = {}
        o = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)       # guard rows behind the output
        ops.gemm(A, B, o[:M], bias=bias, alpha=alpha)
        r["bf16+bias"] = o
        o = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        h = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm(A, B, o[:M], bias=bias, act=L.ACT_GELU, aux=h[:M], alpha=alpha)
        r["gelu"], r["preact"] = o, h
        o = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        cs = torch.full((N,), 0.25, device="cuda")
        ops.gemm(A, B, o[:M], act=L.ACT_GELU_BWD, aux=hpre, alpha=alpha, col_sum=cs)
        r["gelu_bwd"], r["col_sum"] = o, cs
        x = torch.full((M + 3, N), 7.0, device="cuda")
        x[:M] = res
        y = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        ops.gemm(A, B, x[:M], bias=bias, col_scale=gamma, row_scale=rs, residual=x[:M], aux=y[:M], alpha=alpha)  # in place
        r["residual stream"], r["branch"] = x, y
        x = torch.full((M + 3, N), 7.0, device="cuda")
        ops.gemm(A, B, x[:M], bias=bias, residual=res, alpha=alpha)
        r["f32 residual"] = x
        x = torch.full((M + 3, N), 7.0, device="cuda")
        ops.gemm(A, B, x[:M], alpha=alpha)
        r["f32 plain"] = x
        return r

    big, small = run(2), run(0)
    hh = hpre.float().requires_grad_(True)
    torch.nn.functional.gelu(hh).backward(ref)
    want = {"bf16+bias": ref + bias, "gelu": torch.nn.functional.gelu(ref + bias), "preact": ref + bias, "gelu_bwd": hh.grad,
            "residual stream": res + rs[:, None] * gamma[None] * (ref + bias), "branch": ref + bias,
            "f32 residual": res + ref + bias, "f32 plain": ref}
    for k, w in want.items():
        tol = (1e-2, 2e-2) if big[k].dtype == torch.bfloat16 else (1e-3, 5e-3)
        assert_close(big[k][:M], w, tol[0], tol[1], "256-tile " + k)
        assert float((big[k][M:].float() - 7.0).abs().max()) == 0.0, "rows behind M were written: " + k
        # same operation order in both epilogues: the two kernels differ only by the K-order of the fp32 accumulation
        assert_close(big[k][:M], small[k][:M].float(), tol[0], tol[1], "256-tile vs 128-tile " + k)
    assert_close(big["col_sum"], hh.grad.sum(0) + 0.25, 2e-3, 3e-2, "256-tile col_sum")
    assert_close(big["col_sum"], small["col_sum"], 1e-3, 1e-2, "col_sum 256 vs 128")


def test_gemm_big_tile_col_sum_workspace(ops, L, big_tile):
    """Column sums of the 256x256 kernel through the fold workspace: a wave covers a 128-row half alone and stores slot
    2 tm + wm; the ragged last half adds directly."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(77)
    M, N, K = 256 * 4 + 128 + 50, 512, 256
    A = bf(torch.randn(M, K, device="cuda", generator=gen))
    B = bf(torch.randn(N, K, device="cuda", generator=gen))
    h = bf(torch.randn(M, N, device="cuda", generator=gen))
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fold = ops.FoldBatch(A.device, N)
    got = torch.full((N,), 1.5, device="cuda")
    ops.gemm(A, B, out, act=L.ACT_GELU_BWD, aux=h, alpha=0.1, col_sum=got, col_sum_fold=fold)
    assert len(fold.jobs) == 1 and fold.jobs[0][1] == M // 128
    fold.flush()
    hh = h.float().requires_grad_(True)
    torch.nn.functional.gelu(hh).backward((A.float() @ B.float().t()) * 0.1)
    assert_close(got, hh.grad.sum(0) + 1.5, 2e-3, 2e-2, "256-tile col_sum through the fold workspace")
    assert_close(out, hh.grad, 1e-2, 2e-2, "256-tile gelu_bwd")


@pytest.mark.parametrize("M,N,K", [(3072, 768, 13574), (768, 3072, 5000), (768, 768, 2048 + 8), (2304, 768, 54296), (1000, 256, 4100),
                                   (512, 384, 4096), (304, 1152, 2560)])  # N % 256 == 128: the 128x128 split-K path
@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_big_tile_wgrad(ops, L, big_tile, M, N, K, accumulate):
    """dW = A^T B with both operands K-strided through the 256x256 kernel: K slices (ragged K: 13574 = 424*32 + 6) stored
    to the workspace and added by the reduce launch -- against fp32 matmul and against the atomic split-K path."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(M + N + K)
    A = bf(torch.randn(K, M, device="cuda", generator=gen))
    B = bf(torch.randn(K, N, device="cuda", generator=gen))
    base = torch.randn(M, N, device="cuda", generator=gen)
    want = A.float().t() @ B.float() * 0.01 + (base if accumulate else 0)
    got = {}
    for mode in (2, 0):
        L.check(big_tile.vlm_gemm_set_big_tile_mode(mode), "mode")
        c = torch.full((M + 2, N), 3.0, device="cuda")
        c[:M] = base
        ops.gemm(A, B, c[:M], ta=True, tb=True, alpha=0.01, accumulate=accumulate)
        got[mode] = c
    tol = 2e-3 * math.sqrt(K) * 0.01
    assert_close(got[2][:M], want, 1e-3, tol, "256-tile wgrad")
    assert float((got[2][M:] - 3.0).abs().max()) == 0.0
    assert_close(got[2][:M], got[0][:M], 1e-3, tol, "256-tile wgrad vs atomic split-K")
    # the workspace path is deterministic (fixed slice order), the atomic one is not required to be
    L.check(big_tile.vlm_gemm_set_big_tile_mode(2), "mode")
    c2 = torch.full((M + 2, N), 3.0, device="cuda")
    c2[:M] = base
    ops.gemm(A, B, c2[:M], ta=True, tb=True, alpha=0.01, accumulate=accumulate)
    if N % 256 == 0:  # N % 256 == 128 stays on the atomic split-K kernel, which is not required to be deterministic
        assert torch.equal(c2, got[2])


@pytest.mark.parametrize("M,N,K", [(54296, 768, 768), (54296, 2304, 768), (54296, 768, 3072)])
def test_gemm_tail_split_every_operand_offset(ops, L, big_tile, M, N, K):
    """Big-tile mode 3 (the tail split, an option) at the 4B-pass shapes: whole rounds of 256x256 tiles on the big kernel, the remaining ROWS on the
    128x128 kernel (639 tiles = 2.5 rounds -> 43 520 + 10 776 rows).  Every per-row operand of the second call is an
    offset of the first call's: A, C, in-place fp32 residual, row scale, saved pre-activation (output and GELU' input)."""
    L.check(big_tile.vlm_gemm_set_big_tile_mode(3), "mode")
    gen = torch.Generator(device="cuda"); gen.manual_seed(M + N + K)
    A = bf(torch.randn(M, K, device="cuda", generator=gen))
    B = bf(torch.randn(N, K, device="cuda", generator=gen))
    alpha = 1.0 / math.sqrt(K)
    ref = (A.float() @ B.float().t()) * alpha
    bias = torch.randn(N, device="cuda", generator=gen)
    gamma = torch.randn(N, device="cuda", generator=gen) * 0.5
    rs = (torch.rand(M, device="cuda", generator=gen) > 0.3).float() / 0.7
    res = torch.randn(M, N, device="cuda", generator=gen)
    x = torch.full((M + 3, N), 7.0, device="cuda")
    x[:M] = res
    y = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, x[:M], bias=bias, col_scale=gamma, row_scale=rs, residual=x[:M], aux=y[:M], alpha=alpha)  # in place
    assert_close(x[:M], res + rs[:, None] * gamma[None] * (ref + bias), 1e-3, 5e-3, "residual stream")
    assert_close(y[:M], ref + bias, 1e-2, 2e-2, "branch copy")
    assert float((x[M:] - 7.0).abs().max()) == 0.0 and float((y[M:].float() - 7.0).abs().max()) == 0.0
    hpre = bf(torch.randn(M, N, device="cuda", generator=gen))
    o = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
    ops.gemm(A, B, o[:M], act=L.ACT_GELU_BWD, aux=hpre, alpha=alpha)
    hh = hpre.float().requires_grad_(True)
    torch.nn.functional.gelu(hh).backward(ref)
    assert_close(o[:M], hh.grad, 1e-2, 2e-2, "gelu backward")
    assert float((o[M:].float() - 7.0).abs().max()) == 0.0
    # the rows on either side of the split (43 520 for N = 768, 50 944 for N = 2304) come from different kernels
    cut = {768: 43520, 2304: 50944}[N]
    assert_close(o[cut - 4:cut + 4], hh.grad[cut - 4:cut + 4], 1e-2, 2e-2, "rows around the split")


def test_embedding_bwd_matches_torch(ops):
    """vlm_embedding_bwd against torch's embedding backward (padding index skipped, repeated ids summed, accumulation)."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    V, D, n = 1000, 768, 4400
    ids = torch.randint(0, V, (n,), device="cuda", generator=gen)
    ids[::7] = 0          # padding tokens
    ids[1::5] = 17        # a heavily repeated id
    gy = torch.randn(n, D, device="cuda", generator=gen)
    w = torch.zeros(V, D, device="cuda", requires_grad=True)
    torch.nn.functional.embedding(ids, w, padding_idx=0).backward(gy)
    got = torch.full((V, D), 0.25, device="cuda")
    ops.embedding_bwd(gy, ids, got, padding_idx=0)
    assert_close(got - 0.25, w.grad, 1e-5, 1e-4, "embedding backward")
    assert float((got[0] - 0.25).abs().max()) == 0.0  # the padding row receives nothing
    # ids outside [0, vocab) are skipped, never written out of bounds (guard rows behind the table stay untouched)
    table = torch.full((V + 4, D), 0.5, device="cuda")
    bad = ids.clone()
    bad[3::11] = V + 2
    bad[4::13] = -5
    ops.embedding_bwd(gy, bad, table[:V], padding_idx=0)
    keep = (bad >= 0) & (bad < V)
    w2 = torch.zeros(V, D, device="cuda", requires_grad=True)
    torch.nn.functional.embedding(bad[keep], w2, padding_idx=0).backward(gy[keep])
    assert_close(table[:V] - 0.5, w2.grad, 1e-5, 1e-4, "embedding backward with out-of-range ids")
    assert float((table[V:] - 0.5).abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K,mode", [(13574, 3072, 768, 1), (13574, 3072, 768, 0), (880, 3072, 768, 1), (333, 1000, 128, 1)])
def test_gemm_gelu_saved_derivative(ops, L, big_tile, M, N, K, mode):
    """VLM_ACT_GELU_DERIV / VLM_ACT_MUL_AUX: the forward epilogue returns gelu(v) and saves bf16(gelu'(v)); the backward
    epilogue multiplies by it.  Same outputs as ACT_GELU, and the same dh as ACT_GELU_BWD on the saved pre-activation up to
    the bf16 rounding of either saved tensor -- on both kernels (mode 1: by shape, 0: 128x128 always) and a ragged N."""
    L.check(big_tile.vlm_gemm_set_big_tile_mode(mode), "mode")
    gen = torch.Generator(device="cuda"); gen.manual_seed(M + N + K)
    A = bf(torch.randn(M, K, device="cuda", generator=gen))
    B = bf(torch.randn(N, K, device="cuda", generator=gen))
    bias = torch.randn(N, device="cuda", generator=gen) * 0.1
    alpha = 2.0 / math.sqrt(K)
    ref = (A.float() @ B.float().t()) * alpha + bias
    Np = (N + 7) // 8 * 8
    a0 = torch.empty(M, Np, device="cuda", dtype=torch.bfloat16)[:, :N]
    h0 = torch.empty(M, Np, device="cuda", dtype=torch.bfloat16)[:, :N]
    a1 = torch.empty(M, Np, device="cuda", dtype=torch.bfloat16)[:, :N]
    d1 = torch.empty(M, Np, device="cuda", dtype=torch.bfloat16)[:, :N]
    ops.gemm(A, B, a0, bias=bias, act=L.ACT_GELU, aux=h0, alpha=alpha)
    ops.gemm(A, B, a1, bias=bias, act=L.ACT_GELU_DERIV, aux=d1, alpha=alpha)
    assert torch.equal(a0, a1)                                   # the same gelu, bit for bit
    x = ref.clone().requires_grad_(True)
    torch.nn.functional.gelu(x).backward(torch.ones_like(x))
    assert_close(d1, x.grad, 1e-2, 1e-2, "saved gelu'")            # bf16 of the exact derivative
    # backward: dh = (dY W) * gelu'  -- from the saved derivative and from the saved pre-activation
    if N % 8 == 0:
        dy = bf(torch.randn(M, K, device="cuda", generator=gen))   # reuse shapes: "dY" [M,K] x "W^T" [N,K]
        o_old = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        o_new = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ops.gemm(dy, B, o_old, act=L.ACT_GELU_BWD, aux=h0, alpha=alpha)
        ops.gemm(dy, B, o_new, act=L.ACT_MUL_AUX, aux=d1, alpha=alpha)
        want = (dy.float() @ B.float().t()) * alpha * x.grad
        assert_close(o_new, want, 1e-2, 2e-2, "dh from the saved derivative")
        assert_close(o_old, want, 1e-2, 2e-2, "dh from the saved pre-activation")


@pytest.mark.parametrize("rows,N,K", [((3520, 50776 // 8), 768, 768),      # text + image expert rows (image cut for test time)
                                       ((880, 12694), 2304, 768),           # the 22-sample unimodal pair pass
                                       ((256 * 2 + 9, 256 * 3 + 130), 256, 128),
                                       ((40, 577), 512, 256),               # one sample: both groups ragged, one tile each
                                       ((0, 700), 768, 128), ((700, 0), 768, 128),  # an empty expert on either side
                                       ((300, 200, 100, 450), 256, 256),    # four groups
                                       ((333, 444), 192, 192)])             # N % 256 != 0: served as one plain call per group
def test_gemm_grouped_every_epilogue_variant(ops, L, big_tile, rows, N, K):
    """vlm_gemm_bf16_grouped (the experts of an all_moe block in one launch, vision_transformer.py:607-681) against
    (a) fp32 matmul of the bf16 operands per group and (b) one vlm_gemm_bf16 call per group: every epilogue variant a block
    uses, rows of the NEXT group and guard rows behind the last one untouched by a group's ragged last tile."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(sum(rows) + N + K)
    M = sum(rows)
    bounds, r = [], 0
    for n in rows:
        bounds.append((r, r + n)); r += n
    A = bf(torch.randn(M, K, device="cuda", generator=gen))
    Ws = [bf(torch.randn(N, K, device="cuda", generator=gen)) for _ in rows]
    biases = [torch.randn(N, device="cuda", generator=gen) for _ in rows]
    alpha = 1.0 / math.sqrt(K)
    ref = torch.cat([(A[r0:r1].float() @ W.float().t()) * alpha + b for (r0, r1), W, b in zip(bounds, Ws, biases)])
    gamma = torch.randn(N, device="cuda", generator=gen) * 0.5
    rs = (torch.rand(M, device="cuda", generator=gen) > 0.3).float() / 0.7
    res = torch.randn(M, N, device="cuda", generator=gen)
    hpre = bf(torch.randn(M, N, device="cuda", generator=gen))

    def groups(with_bias=True, col_sums=None):
        return [(r0, r1, W, b if with_bias else None, col_sums[i] if col_sums else None)
                for i, ((r0, r1), W, b) in enumerate(zip(bounds, Ws, biases))]

    def run(grouped):
        def call(a, out, with_bias=True, col_sums=None, **kw):
            if grouped:
                return ops.gemm_grouped(a, groups(with_bias, col_sums), out, alpha=alpha, **kw)
            for i, (r0, r1, W, b, cs) in enumerate(groups(with_bias, col_sums)):
                if r1 > r0:
                    sl = {k: (v[r0:r1] if k in ("aux", "row_scale", "residual") and v is not None else v) for k, v in kw.items()}
                    ops.gemm(a[r0:r1], W, out[r0:r1], bias=b, col_sum=cs, alpha=alpha, **sl)
        r = {}
        o = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        call(A, o[:M])
        r["bf16+bias"] = o
        o = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        h = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        call(A, o[:M], act=L.ACT_GELU_DERIV, aux=h[:M])
        r["gelu"], r["gelu'"] = o, h
        o = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        css = [torch.full((N,), 0.25 * (i + 1), device="cuda") for i in range(len(rows))]
        call(A, o[:M], with_bias=False, col_sums=css, act=L.ACT_MUL_AUX, aux=hpre)
        r["mul_aux"], r["col_sums"] = o, css
        x = torch.full((M + 3, N), 7.0, device="cuda")
        x[:M] = res
        y = torch.full((M + 3, N), 7.0, device="cuda", dtype=torch.bfloat16)
        call(A, x[:M], col_scale=gamma, row_scale=rs, residual=x[:M], aux=y[:M])  # in place
        r["residual stream"], r["branch"] = x, y
        return r

    g, s = run(True), run(False)
    pre = ref.clone().requires_grad_(True)
    act = torch.nn.functional.gelu(pre)
    dact = torch.autograd.grad(act.sum(), pre)[0]
    nob = ref - torch.cat([b[None].expand(r1 - r0, N) for (r0, r1), b in zip(bounds, biases)])
    want = {"bf16+bias": ref, "gelu": act.detach(), "gelu'": dact, "mul_aux": nob * hpre.float(),
            "residual stream": res + rs[:, None] * gamma[None] * ref, "branch": ref}
    for k, w in want.items():
        tol = (1e-2, 2e-2) if g[k].dtype == torch.bfloat16 else (1e-3, 5e-3)
        assert_close(g[k][:M], w, tol[0], tol[1], "grouped " + k)
        assert float((g[k][M:].float() - 7.0).abs().max()) == 0.0, "rows behind M were written: " + k
        assert_close(g[k][:M], s[k][:M].float(), tol[0], tol[1], "grouped vs per-group " + k)
    for i, (r0, r1) in enumerate(bounds):
        wcs = (nob[r0:r1] * hpre[r0:r1].float()).sum(0) + 0.25 * (i + 1)
        assert_close(g["col_sums"][i], wcs, 2e-3, 3e-2, "grouped col_sum of group %d" % i)
        assert_close(g["col_sums"][i], s["col_sums"][i], 1e-3, 1e-2, "col_sum grouped vs per-group %d" % i)


def test_gemm_grouped_col_sum_fold_and_group_isolation(ops, L, big_tile):
    """Column sums of a grouped call through the fold workspace (a region per group, rows counted from the group's first
    row), and the row bound between groups: a group's ragged last tile must neither read its operands from nor write into
    the next group's rows (checked with a poisoned second group)."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    N, K = 512, 256
    bounds = [(0, 256 + 128 + 50), (256 + 128 + 50, 256 + 128 + 50 + 256 * 2 + 128 + 7)]
    M = bounds[-1][1]
    A = bf(torch.randn(M, K, device="cuda", generator=gen))
    Ws = [bf(torch.randn(N, K, device="cuda", generator=gen)) for _ in bounds]
    h = bf(torch.randn(M, N, device="cuda", generator=gen))
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    fold = ops.FoldBatch(A.device, N)
    got = [torch.full((N,), 1.5, device="cuda"), torch.full((N,), -2.0, device="cuda")]
    ops.gemm_grouped(A, [(r0, r1, W, None, cs) for (r0, r1), W, cs in zip(bounds, Ws, got)], out, act=L.ACT_MUL_AUX, aux=h,
                     alpha=0.1, col_sum_fold=fold)
    assert [j[1] for j in fold.jobs] == [(r1 - r0) // 128 for r0, r1 in bounds]
    fold.flush()
    for (r0, r1), W, cs, base in zip(bounds, Ws, got, (1.5, -2.0)):
        w = (A[r0:r1].float() @ W.float().t()) * 0.1 * h[r0:r1].float()
        assert_close(out[r0:r1], w, 1e-2, 2e-2, "grouped mul_aux rows %d:%d" % (r0, r1))
        assert_close(cs, w.sum(0) + base, 2e-3, 2e-2, "grouped col_sum through the fold workspace")
    # NaN in group 1's operand rows and weight: group 0's outputs and column sums must not see them
    A2 = A.clone(); A2[bounds[1][0]:] = float("nan")
    out2 = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    cs0 = torch.zeros(N, device="cuda")
    ops.gemm_grouped(A2, [(bounds[0][0], bounds[0][1], Ws[0], None, cs0), (bounds[1][0], bounds[1][1], Ws[1] * float("nan"), None, None)],
                     out2, act=L.ACT_MUL_AUX, aux=h, alpha=0.1)
    assert torch.isfinite(out2[:bounds[0][1]].float()).all() and torch.isfinite(cs0).all()
    assert torch.equal(out2[:bounds[0][1]], out[:bounds[0][1]])


def test_gemm_grouped_argument_errors(ops, L):
    A = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    W = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    out = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(L.VlmError):
        ops.gemm_grouped(A, [(32, 64, W, None, None), (0, 32, W, None, None)], out)  # not ascending
    with pytest.raises(L.VlmError):
        ops.gemm_grouped(A, [(0, 40, W, None, None), (32, 64, W, None, None)], out)  # overlapping
    with pytest.raises(L.VlmError):
        ops.gemm_grouped(A, [(0, 16, W, None, None)] * 5, out)                       # too many groups
    ops.gemm_grouped(A, [(0, 64, W, None, None)], out)


@pytest.mark.parametrize("rows,M,N", [((3520, 6000), 768, 768), ((880, 12694), 2304, 768), ((40, 577), 256, 512),
                                       ((0, 900), 768, 256), ((900, 0), 768, 256), ((130, 70, 333, 64), 256, 256),
                                       ((500, 700), 192, 192)])  # N % 256 != 0: one plain call per group
@pytest.mark.parametrize("accumulate", [False, True])
def test_gemm_wgrad_grouped(ops, L, big_tile, rows, M, N, accumulate):
    """vlm_gemm_wgrad_grouped: dW_g (+)= dY[rows_g]^T X[rows_g] for the experts of a block in one launch, against fp32 matmul
    per group and against one wgrad call per group; ragged token counts (not multiples of 32), empty groups."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(sum(rows) + M + N)
    T = sum(rows)
    bounds, r = [], 0
    for n in rows:
        bounds.append((r, r + n)); r += n
    dy = bf(torch.randn(T, M, device="cuda", generator=gen))
    x = bf(torch.randn(T, N, device="cuda", generator=gen))
    base = [torch.randn(M, N, device="cuda", generator=gen) for _ in rows]
    got = [b.clone() for b in base]
    ops.gemm_wgrad_grouped(dy, x, [(r0, r1, g) for (r0, r1), g in zip(bounds, got)], accumulate=accumulate)
    for (r0, r1), g, b in zip(bounds, got, base):
        want = dy[r0:r1].float().t() @ x[r0:r1].float() + (b if accumulate else 0)
        tol = 2e-3 * math.sqrt(max(r1 - r0, 1))
        assert_close(g, want, 1e-3, tol, "grouped wgrad rows %d:%d" % (r0, r1))
        if r1 > r0:
            one = b.clone()
            ops.gemm(dy[r0:r1], x[r0:r1], one, ta=True, tb=True, accumulate=accumulate)
            assert_close(g, one, 1e-3, tol, "grouped vs plain wgrad rows %d:%d" % (r0, r1))


@pytest.mark.parametrize("rows,V,frac_ignored", [(880, 30522, 0.8), (37, 1024, 0.5), (5, 30522, 1.0), (64, 777, 0.0)])
def test_cross_entropy_matches_torch(pkg, ops, rows, V, frac_ignored):
    """vlm_cross_entropy_fwd / _bwd (the MLM head's loss, objectives.py:88-143) against F.cross_entropy on the fp32 upcast of the same
    bf16 logits: loss, the gradient (as the zero-padded bf16 matrix the decoder GEMMs take), ignore_index rows, a vocabulary that is
    not a multiple of 8, every row ignored (nan, like torch)."""
    engine = importlib.import_module("vl_merging_amd.engine")
    gen = torch.Generator(device="cuda"); gen.manual_seed(rows + V)
    Vp = (V + 63) // 64 * 64
    buf = torch.full((rows, Vp), 9.0, device="cuda", dtype=torch.bfloat16)  # what the decoder GEMM leaves: garbage in the padding
    buf[:, :V] = (torch.randn(rows, V, device="cuda", generator=gen) * 3).to(torch.bfloat16)
    logits = buf[:, :V].detach().requires_grad_(True)
    labels = torch.randint(0, V, (rows,), device="cuda", generator=gen)
    labels[torch.rand(rows, device="cuda", generator=gen) < frac_ignored] = -100
    if frac_ignored == 1.0:
        labels[:] = -100
    loss = engine.cross_entropy(logits, labels, ignore_index=-100)
    ref_in = logits.detach().float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, labels, ignore_index=-100)
    if frac_ignored == 1.0:
        assert torch.isnan(loss) and torch.isnan(ref)
        return
    assert abs(float(loss) - float(ref)) <= 2e-5 * max(1.0, abs(float(ref)))
    (loss * 1.7).backward()
    (ref * 1.7).backward()
    g = logits.grad
    assert g.dtype == torch.bfloat16
    assert_close(g, ref_in.grad, 1e-2, 1e-7, "cross-entropy gradient")
    assert float(g[labels == -100].abs().max() if (labels == -100).any() else 0.0) == 0.0


def test_l2_normalize_matches_autograd(pkg):
    engine = importlib.import_module("vl_merging_amd.engine")
    gen = torch.Generator(device="cuda"); gen.manual_seed(3)
    x = torch.randn(22, 768, device="cuda", generator=gen).to(torch.bfloat16)
    a = x.clone().requires_grad_(True)
    b = x.clone().requires_grad_(True)
    ya = engine.l2_normalize(a)
    bf = b.float()
    yb = bf / bf.norm(dim=-1, keepdim=True)
    w = torch.randn(22, 768, device="cuda", generator=gen)
    (ya * w).sum().backward()
    (yb * w).sum().backward()
    assert_close(ya, yb, 1e-6, 1e-7, "l2 forward")
    assert_close(a.grad, b.grad, 1e-2, 1e-4, "l2 backward")


def test_layernorm_bwd_scale_with_a_frozen_layernorm_keeps_the_layerscale_sums(ops):
    """A frozen LayerNorm (no dgamma / dbeta) in front of a trainable LayerScale: the fused call still parks the LayerScale's
    partials as a fold job (the library reports the grid whenever either set was written), and the next reservation of the
    batch does not overlap the region they sit in."""
    M, D = 1000, 192
    gen = torch.Generator(device="cuda"); gen.manual_seed(11)
    x = torch.randn(M, D, device="cuda", generator=gen)
    g = torch.ones(D, device="cuda")
    stats = torch.empty(M, 2, device="cuda")
    ops.layernorm_fwd(x, g, torch.zeros(D, device="cuda"), 1e-6, torch.empty(M, D, device="cuda", dtype=torch.bfloat16), stats)
    dy = bf(torch.randn(M, D, device="cuda", generator=gen))
    y = bf(torch.randn(M, D, device="cuda", generator=gen))
    sg = torch.randn(D, device="cuda", generator=gen) * 0.1

    def run(fused):
        dx = torch.empty(M, D, device="cuda"); sdy = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
        dsg, dsb = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        other = [torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")]
        fb = ops.FoldBatch(x.device, D)
        if fused:
            ops.layernorm_bwd_scale(dy, x, stats, g, dx, None, None, None, y=y, sgamma=sg, row_scale=None, sdy=sdy,
                                    dsgamma=dsg, dsbias=dsb, fold=fb)
        else:
            ops.layernorm_bwd(dy, x, stats, g, dx, fold=fb)
            ops.layerscale_bwd(dx, y, sg, None, sdy, dsg, dsb, fold=fb)
        # a further job of the same batch must land in a region of its own
        ops.layernorm_bwd(dy, x, stats, g, torch.empty(M, D, device="cuda"), dgamma=other[0], dbeta=other[1], fold=fb)
        fb.flush()
        torch.cuda.synchronize()
        return dx, sdy, dsg, dsb, other

    a, b = run(False), run(True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert float(b[2].abs().max()) > 0 and float(b[3].abs().max()) > 0
    assert_close(b[2], a[2], 1e-5, 1e-4 * math.sqrt(M), "dsgamma")
    assert_close(b[3], a[3], 1e-5, 1e-4 * math.sqrt(M), "dsbias")
    for u, v in zip(a[4], b[4]):
        assert_close(v, u, 1e-5, 1e-4 * math.sqrt(M), "the following job's sums")


def test_cross_entropy_counts_only_the_rows_the_kernels_score(pkg, ops):
    """A label outside [0, V) (which F.cross_entropy rejects with a device assert) gets neither loss nor gradient from the row
    kernels; it must not inflate the mean's denominator either."""
    engine = importlib.import_module("vl_merging_amd.engine")
    gen = torch.Generator(device="cuda"); gen.manual_seed(5)
    rows, V = 16, 200
    buf = torch.zeros(rows, 256, device="cuda", dtype=torch.bfloat16)
    buf[:, :V] = torch.randn(rows, V, device="cuda", generator=gen).to(torch.bfloat16)
    logits = buf[:, :V]
    labels = torch.randint(0, V, (rows,), device="cuda", generator=gen)
    good = engine.cross_entropy(logits[:12], labels[:12])
    labels_bad = labels.clone()
    labels_bad[12:] = torch.tensor([V, V + 7, -5, 100000], device="cuda")
    mixed = engine.cross_entropy(logits, labels_bad)
    assert abs(float(mixed) - float(good)) <= 1e-6 * max(1.0, abs(float(good)))


@pytest.mark.parametrize("M,N,K,with_rs", [(700, 192, 192, True), (1300, 768, 3072, False), (300, 768, 768, True)])
def test_layerscale_fold_and_finish_match_autograd(pkg, ops, M, N, K, with_rs):
    """LayerScale folded into the branch's output projection (csrc/layerscale.hip; vision_transformer.py:489-491, :586, :603):
    forward x + rs * (a W'^T + b') with W' = diag(gamma) W, b' = gamma * b; backward from the RAW sums G = g^T a, s = colsum(g),
    g = bf16(rs * dx):  dW = gamma G,  db = gamma s,  dgamma = sum_k W G + b s  -- against torch autograd of the unfolded
    x + rs * gamma * (a W^T + b) in fp32 (tolerances: the operands are bf16 in the kernels)."""
    gen = torch.Generator(device="cuda"); gen.manual_seed(M + N)
    W = torch.randn(N, K, device="cuda", generator=gen) * 0.05
    b = torch.randn(N, device="cuda", generator=gen) * 0.1
    gamma = 0.1 + 0.05 * torch.randn(N, device="cuda", generator=gen)
    a = bf(torch.randn(M, K, device="cuda", generator=gen))
    x = torch.randn(M, N, device="cuda", generator=gen)
    dx = torch.randn(M, N, device="cuda", generator=gen)
    rs = ((torch.rand(M, device="cuda", generator=gen) > 0.3).float() / 0.7) if with_rs else None
    # ---- reference
    Wr, br, gr = W.clone().requires_grad_(True), b.clone().requires_grad_(True), gamma.clone().requires_grad_(True)
    y = a.float() @ Wr.t() + br
    out_ref = x + (rs[:, None] if rs is not None else 1.0) * gr * y
    (out_ref * dx).sum().backward()
    # ---- folded path
    shadow = torch.empty(N, K, device="cuda", dtype=torch.bfloat16)
    bfold = torch.empty(N, device="cuda")
    ops.layerscale_fold([dict(weight=W, gamma=gamma, bias=b, shadow=shadow, bias_out=bfold)])
    assert torch.equal(shadow, (gamma[:, None] * W).to(torch.bfloat16)) and torch.equal(bfold, gamma * b)
    out = torch.empty(M, N, device="cuda")
    ops.gemm(a, shadow, out, bias=bfold, row_scale=rs, residual=x)
    assert_close(out, out_ref.detach(), 2e-2, 2e-2 * float((out_ref.detach() - x).abs().max()), "folded forward")
    g = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    raw_w = torch.zeros(N, K, device="cuda"); raw_b = torch.zeros(N, device="cuda")
    ops.layerscale_bwd(dx, None, None, rs, g, None, raw_b)
    want_g = ((rs[:, None] if rs is not None else 1.0) * dx).to(torch.bfloat16)
    assert torch.equal(g, want_g)
    ops.gemm(g, a, raw_w, ta=True, tb=True, accumulate=True)
    dW = torch.full((N, K), 0.25, device="cuda"); db = torch.full((N,), -0.5, device="cuda"); dg = torch.full((N,), 2.0, device="cuda")
    ops.layerscale_finish([dict(weight=W, gamma=gamma, bias=b, raw_w=raw_w, raw_b=raw_b, dweight=dW, dbias=db, dgamma=dg)])
    torch.cuda.synchronize()
    assert float(raw_w.abs().max()) == 0.0 and float(raw_b.abs().max()) == 0.0  # zeroed: composes with accumulation
    assert_close(dW - 0.25, Wr.grad, 2e-2, 1e-2 * float(Wr.grad.abs().max()), "dW")
    assert_close(db + 0.5, br.grad, 2e-2, 1e-2 * float(br.grad.abs().max()), "db")
    assert_close(dg - 2.0, gr.grad, 2e-2, 1e-2 * float(gr.grad.abs().max()), "dgamma")


# ---- round 6: the loss tail (csrc/lossops.hip) against torch ------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,D", [(22, 768), (3, 192), (1, 64), (67, 200)])
def test_l2norm_fwd_bwd_match_autograd(ops, rows, D, dtype):
    g = torch.Generator(device="cuda"); g.manual_seed(rows * 7 + D)
    x = (torch.randn(rows, D, device="cuda", generator=g) * 3).to(dtype)
    y, inv = ops.l2norm_fwd(x)
    xr = x.float().requires_grad_(True)
    yr = xr / xr.norm(dim=-1, keepdim=True)
    assert_close(y, yr.detach(), 1e-6, 1e-6, "l2norm y")
    gy = torch.randn(rows, D, device="cuda", generator=g)
    (gx,) = torch.autograd.grad(yr, xr, gy)
    dx = ops.l2norm_bwd(gy.contiguous(), y, inv, dtype)
    assert dx.dtype == dtype
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert_close(dx.float(), gx, tol, tol * float(gx.abs().max()), "l2norm dx")


@pytest.mark.parametrize("B,n,D", [(22, 22, 768), (3, 3, 192), (2, 6, 64), (22, 176, 768), (5, 20, 200)])
def test_contrastive_loss_and_gradients_match_torch(ops, B, n, D):
    """vlm_contrastive against the reference's formulation (objectives.py:274-300): gathered features with this rank's B rows
    first, gradients through the own rows only, the logit scale's gradient from every entry."""
    g = torch.Generator(device="cuda"); g.manual_seed(B * 1000 + n)
    img = torch.nn.functional.normalize(torch.randn(n, D, device="cuda", generator=g), dim=-1).contiguous()
    txt = torch.nn.functional.normalize(torch.randn(n, D, device="cuda", generator=g), dim=-1).contiguous()
    ls = torch.tensor([2.3], device="cuda")
    out3, logits, d_img, d_txt = ops.contrastive(img, txt, B, ls)
    io = img[:B].clone().requires_grad_(True); to = txt[:B].clone().requires_grad_(True); lsr = ls.clone().requires_grad_(True)
    ai = torch.cat([io, img[B:]]); at = torch.cat([to, txt[B:]])
    li = lsr.exp() * ai @ at.t()
    gt = torch.arange(n, device="cuda")
    loss = (torch.nn.functional.cross_entropy(li, gt) + torch.nn.functional.cross_entropy(li.t(), gt)) / 2
    loss.backward()
    assert_close(logits, li.detach(), 1e-5, 1e-5, "logits")
    assert abs(float(out3[0]) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    assert abs(float(out3[2]) - float(ls.exp())) <= 1e-5 * float(ls.exp())
    assert abs(float(out3[1]) - float(lsr.grad)) <= 1e-4 * max(1.0, abs(float(lsr.grad)))
    assert_close(d_img, io.grad, 1e-4, 1e-6, "d img")
    assert_close(d_txt, to.grad, 1e-4, 1e-6, "d txt")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,V", [(66, 2), (6, 2), (300, 5), (1, 3)])
def test_small_cross_entropy_matches_torch(ops, rows, V, dtype):
    g = torch.Generator(device="cuda"); g.manual_seed(rows + V)
    buf = (torch.randn(rows, 64, device="cuda", generator=g) * 2).to(dtype)
    logits = buf[:, :V]  # a view with a row stride, like the ITM head's padded output
    labels = torch.randint(0, V, (rows,), device="cuda", generator=g)
    loss, d = ops.small_cross_entropy(logits, labels)
    lr = logits.float().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lr, labels)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 1e-5 * max(1.0, abs(float(ref)))
    assert_close(d, lr.grad, 1e-5, 1e-6, "small ce grad")


def test_cross_entropy_reduce_and_scale_by_scalar(ops):
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    rows, V = 880, 30522
    loss_rows = torch.rand(rows, device="cuda", generator=g)
    labels = torch.randint(0, V, (rows,), device="cuda", generator=g)
    labels[::3] = -100
    labels[5] = V + 3  # outside the vocabulary: not counted either
    out2 = ops.cross_entropy_reduce(loss_rows, labels, V, -100)
    keep = (labels != -100) & (labels >= 0) & (labels < V)
    assert abs(float(out2[0]) - float(loss_rows[keep].sum() / keep.sum())) <= 1e-5
    assert abs(float(out2[1]) - 1.0 / float(keep.sum())) <= 1e-9
    a, b, c = torch.randn(22, 768, device="cuda", generator=g), torch.randn(7, device="cuda", generator=g), torch.randn(1, device="cuda", generator=g)
    sc = torch.tensor([0.37], device="cuda")
    oa, ob, oc = ops.scale_by_scalar([a, b, c], sc)
    assert torch.equal(oa, a * sc) and torch.equal(ob, b * sc) and torch.equal(oc, c * sc)


# ---- round 6: the two ends of a pass (csrc/frontops.hip) against torch --------------------------------------------------------
@pytest.mark.parametrize("B,T,D,V,drop", [(4, 40, 768, 3000, True), (3, 7, 192, 50, False), (22, 40, 768, 30522, True), (1, 1, 64, 9, True)])
def test_text_rows_fwd_bwd_match_torch(ops, B, T, D, V, drop):
    """vlm_text_rows_fwd / _bwd against BertEmbeddings.forward + the modality type row written with torch ops (reference
    vilt_module.py:51-63, :1111-1113): gather, + bert type 0, LayerNorm(1e-12), dropout with a given keep mask, + vilt type 0."""
    g = torch.Generator(device="cuda"); g.manual_seed(B * 100 + T)
    n = B * T
    ids = torch.randint(0, V, (n,), device="cuda", generator=g)
    ids[::5] = 0  # padding rows: looked up like any row, no gradient
    word = torch.randn(V, D, device="cuda", generator=g)
    bt = torch.randn(D, device="cuda", generator=g) * 0.1
    gamma = 1 + 0.1 * torch.randn(D, device="cuda", generator=g); beta = 0.1 * torch.randn(D, device="cuda", generator=g)
    vt = torch.randn(D, device="cuda", generator=g) * 0.1
    u = torch.rand(n, D, device="cuda", generator=g) if drop else None
    p, scale = (0.1, 1 / 0.9) if drop else (0.0, 1.0)
    x = torch.zeros(n + 5, D, device="cuda")
    stats = ops.text_rows_fwd(ids, word, bt, gamma, beta, 1e-12, x[:n], u, p, scale, add1=vt)
    wr, btr, gr, br, vtr = [t.clone().requires_grad_(True) for t in (word, bt, gamma, beta, vt)]
    e = torch.nn.functional.embedding(ids, wr, padding_idx=0) + btr
    y = torch.nn.functional.layer_norm(e, (D,), gr, br, 1e-12)
    if drop:
        y = y * (u >= p).float() * scale
    ref = y + vtr
    assert_close(x[:n], ref.detach(), 1e-5, 2e-5, "text rows")
    assert float(x[n:].abs().max()) == 0.0
    gy = torch.randn(n + 5, D, device="cuda", generator=g)
    ref.backward(gy[:n])
    dword = torch.full((V, D), 0.5, device="cuda"); d1 = torch.full((D,), 1.0, device="cuda"); db = torch.full((D,), 2.0, device="cuda")
    dg = torch.full((D,), 3.0, device="cuda"); d0 = torch.full((D,), 4.0, device="cuda")
    ops.text_rows_bwd(gy[:n], ids, word, bt, gamma, stats, u, p, scale, dword, 0, d1, db, dg, d0)
    torch.cuda.synchronize()
    for got, want, base, what in ((dword, wr.grad, 0.5, "d word"), (d1, vtr.grad, 1.0, "d type row"), (db, br.grad, 2.0, "d beta"),
                                  (dg, gr.grad, 3.0, "d gamma"), (d0, btr.grad, 4.0, "d bert type")):
        assert_close(got - base, want, 2e-4, 2e-4 * float(want.abs().max()) + 1e-5, what)
    assert float((dword[0] - 0.5).abs().max()) == 0.0  # the padding row


@pytest.mark.parametrize("B,rows,D", [(22, 577, 768), (3, 10, 192), (1, 2, 64), (88, 577, 768)])
def test_image_rows_kernels_match_torch(ops, B, rows, D):
    g = torch.Generator(device="cuda"); g.manual_seed(B + rows)
    cb = torch.randn(D, device="cuda", generator=g); tt = torch.randn(D, device="cuda", generator=g); cls = torch.randn(1, 1, D, device="cuda", generator=g)
    pre = ops.image_rows_prep(cb, tt, cls)
    assert torch.equal(pre[0], cb + tt) and torch.equal(pre[1], cls.view(-1) + tt)
    assert torch.equal(ops.image_rows_prep(None, tt, cls)[0], tt)
    x = torch.randn(B * rows + 3, D, device="cuda", generator=g)
    keep = x.clone()
    ops.image_lead_rows(x[3:], B, rows, pre[1])
    want = keep.clone(); want[3:].view(B, rows, D)[:, 0] = pre[1]
    assert torch.equal(x, want)
    gy = torch.randn(B * rows + 3, D, device="cuda", generator=g)
    dbias = torch.full((D,), 1.0, device="cuda"); dtt = torch.full((D,), 2.0, device="cuda"); dcls = torch.full((1, 1, D), 3.0, device="cuda")
    g16 = ops.image_rows_bwd(gy[3:], B, rows, dbias, dtt, dcls)
    gv = gy[3:].view(B, rows, D)
    ref16 = gv.clone(); ref16[:, 0] = 0
    assert torch.equal(g16.view(B, rows, D), ref16.to(torch.bfloat16))
    lead = gv[:, 0].double().sum(0); patch = gv[:, 1:].double().sum((0, 1))
    tol = 1e-5 * float(gv.abs().max()) * (B * rows) ** 0.5 + 1e-6
    assert_close(dbias - 1.0, patch.float(), 1e-5, tol, "d conv bias")
    assert_close(dtt - 2.0, (patch + lead).float(), 1e-5, tol, "d type row")
    assert_close(dcls.view(-1) - 3.0, lead.float(), 1e-5, tol, "d cls")
    g16b = ops.image_rows_bwd(gy[3:], B, rows, None, None, None)  # nothing wanted: the cast alone
    assert torch.equal(g16b, g16)


@pytest.mark.parametrize("M,N,Np", [(66, 2, 64), (22, 768, 768), (880, 768, 768), (5, 30, 64)])
@pytest.mark.parametrize("gdtype", [torch.float32, torch.bfloat16])
def test_head_activation_kernels_match_torch(ops, M, N, Np, gdtype):
    g = torch.Generator(device="cuda"); g.manual_seed(M + N)
    pre = (torch.randn(M, Np, device="cuda", generator=g) * 2).to(torch.bfloat16)
    gy = torch.randn(M, N, device="cuda", generator=g).to(gdtype)
    y = ops.tanh_fwd(pre[:, :N])
    assert_close(y, torch.tanh(pre[:, :N].float()), 1e-6, 1e-6, "tanh")
    h = pre[:, :N].float().requires_grad_(True)
    (gh,) = torch.autograd.grad(torch.nn.functional.gelu(h), h, gy.float())
    for mode, saved, want in ((ops.ACT_BWD_GELU, pre[:, :N], gh), (ops.ACT_BWD_TANH, y, gy.float() * (1 - y * y)), (ops.ACT_BWD_NONE, None, gy.float())):
        dy = ops.act_bwd(gy, saved, mode, Np)
        assert dy.shape == (M, Np) and dy.dtype == torch.bfloat16
        assert_close(dy[:, :N], want, 2 ** -7, 1e-6, "act_bwd mode %d" % mode)
        if Np > N:
            assert float(dy[:, N:].abs().max()) == 0.0
    if N <= 64:
        out = torch.full((N,), 1.5, device="cuda")
        a = gy.to(torch.bfloat16)
        ops.colsum_small(a, out)
        assert_close(out - 1.5, a.float().sum(0), 1e-5, 1e-4, "colsum_small")


def test_sample_negatives_distribution_and_edge_cases(ops):
    """vlm_sample_negatives against F.softmax + fill_diagonal_(0) + multinomial's DISTRIBUTION (objectives.py:197-215): never the
    own index, frequencies within 5 sigma of the weights over 4 000 draws, a forced draw at n = 2, a transposed second matrix."""
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    B, n = 6, 20
    sim = torch.randn(n, n, device="cuda", generator=g) * 2
    a, b = sim[:B], sim.t()[:B]
    draws = 4000
    counts = torch.zeros(2, B, n, device="cuda")
    for _ in range(draws // 50):
        for _ in range(50):
            idx = ops.sample_negatives(a, b, B, torch.rand(2, B, device="cuda", generator=g))
            counts.scatter_add_(2, idx.unsqueeze(-1), torch.ones(2, B, 1, device="cuda"))
    for d, m in ((0, a), (1, b)):
        w = torch.softmax(m.float(), dim=1)
        w[torch.arange(B), torch.arange(B)] = 0
        w = w / w.sum(1, keepdim=True)
        assert float(counts[d][torch.arange(B), torch.arange(B)].max()) == 0.0
        sigma = (w * (1 - w) / draws).sqrt()
        assert bool(((counts[d] / draws - w).abs() <= 5 * sigma + 1e-3).all())
    two = torch.randn(2, 2, device="cuda", generator=g)
    for _ in range(5):
        idx = ops.sample_negatives(two, two.t(), 2, torch.rand(2, 2, device="cuda", generator=g))
        assert idx.tolist() == [[1, 0], [1, 0]]
    for uval in (0.0, 0.99999994):  # the ends of the uniform range pick the first / last candidate with weight
        idx = ops.sample_negatives(a, b, B, torch.full((2, B), uval, device="cuda"))
        assert bool((idx != torch.arange(B, device="cuda")).all()) and bool(((idx >= 0) & (idx < n)).all())


def test_weighted_sum_matches_torch(ops):
    t = [torch.tensor(v, device="cuda") for v in (1.5, -2.25, 8.0)]
    out = ops.weighted_sum(t, [0.5, 2.0, 1.0])
    assert abs(float(out) - (0.75 - 4.5 + 8.0)) < 1e-6 and out.dim() == 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,T,I,D", [(4, 40, 577, 768), (2, 5, 3, 64), (3, 8, 0, 192), (2, 0, 7, 64)])
def test_feature_views_gradient_matches_autograd_views(pkg, ops, B, T, I, D, dtype):
    """engine.feature_views / row_range against the plain views they replace: same values (they ARE views), same gradient of x
    whichever subset of the views is used downstream."""
    engine = importlib.import_module("vl_merging_amd.engine")
    g = torch.Generator(device="cuda"); g.manual_seed(B + T + I)
    x0 = torch.randn(B * T + B * I, D, device="cuda", generator=g).to(dtype)
    w = [torch.randn(s, device="cuda", generator=g).to(dtype) for s in ((B, T, D), (B, I, D), (B, D), (B, D))]
    for used in ((0, 1, 2, 3), (2, 3), (0, 2), (3,), (1,)):
        xa = x0.clone().requires_grad_(True)
        xb = x0.clone().requires_grad_(True)
        va = engine.feature_views(xa, B, T, I)
        nt = B * T
        tb, ib = xb[:nt].view(B, T, D), xb[nt:].view(B, I, D)
        vb = (tb, ib, tb[:, 0] if T else xb[:0], ib[:, 0] if I else xb[:0])
        for k in range(4):
            assert torch.equal(va[k], vb[k])
        la = sum((va[k].float() * w[k].float()).sum() for k in used if va[k].numel())
        lb = sum((vb[k].float() * w[k].float()).sum() for k in used if vb[k].numel())
        if not torch.is_tensor(la):
            continue
        la.backward(); lb.backward()
        tol = 0.0 if dtype == torch.float32 else 2 ** -7
        assert_close(xa.grad, xb.grad, tol, 1e-6, "feature_views grad %s" % (used,))
    ta = x0.view(-1, D)[: B * 2].view(B, 2, D).clone().requires_grad_(True) if (B * T + B * I) >= 2 * B else None
    if ta is not None:
        tb_ = ta.detach().clone().requires_grad_(True)
        ra, rb = engine.row_range(ta, 1, None), tb_[1:]
        assert torch.equal(ra, rb)
        gg = torch.randn(ra.shape, device="cuda", generator=g).to(dtype)
        ra.backward(gg); rb.backward(gg)
        assert torch.equal(ta.grad, tb_.grad)
