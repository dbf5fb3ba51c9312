"""Fused attention HIP kernels vs a plain PyTorch fp32 restatement of reference vision_transformer.py:346-358.

Tolerance (forward): inputs are bf16-exact in both paths; the kernel rounds P to bf16 before P.V and O to bf16:
|err| <= 2e-2 * max|V| absolute + 1e-2 relative covers it (P rows sum to 1).
"""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops(pkg):
    return importlib.import_module("vl_merging_amd.ops")


@pytest.fixture(scope="module")
def L(pkg):
    return importlib.import_module("vl_merging_amd._lib")


def build_case(B, n0, n1, H, seed, with_bias=True, with_mask=True):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    D = H * 64
    rows = B * (n0 + n1)
    qkv = (torch.randn(rows, 3 * D, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
    pos1 = (n0 + 7) // 8 * 8
    NP = pos1 + n1
    ld = (NP + 3) // 4 * 4
    R = 300
    idx = torch.randint(0, R, (NP, ld), device="cuda", generator=g).to(torch.int16)
    # structured parts like the reference: constant text->image / image->text blocks
    if n0 and n1:
        idx[:n0, pos1:] = R - 2
        idx[pos1:, :n0] = R - 1
    table = torch.randn(R, 2 * H, device="cuda", generator=g)  # two "layers"
    keep0 = None
    if with_mask and n0:
        lens = torch.randint(max(1, n0 // 4), n0 + 1, (B,), device="cuda", generator=g)
        keep0 = (torch.arange(n0, device="cuda")[None] < lens[:, None]).to(torch.uint8).contiguous()
    return dict(B=B, n0=n0, n1=n1, H=H, D=D, qkv=qkv, idx=idx, pos1=pos1, table=table if with_bias else None,
                keep0=keep0, R=R)


def make_idx_t(c):
    NP = c["pos1"] + c["n1"]
    idx_t = torch.zeros(NP, (NP + 3) // 4 * 4, device="cuda", dtype=torch.int16)
    idx_t[:, :NP] = c["idx"][:NP, :NP].t() * 4
    return idx_t


def to_seq(x, c):
    """segment-major [rows, F] -> [B, n0+n1, F]"""
    B, n0, n1 = c["B"], c["n0"], c["n1"]
    F = x.shape[-1]
    t = x[: B * n0].view(B, n0, F)
    i = x[B * n0:].view(B, n1, F)
    return torch.cat([t, i], 1)


def from_seq(y, c):
    B, n0, n1 = c["B"], c["n0"], c["n1"]
    F = y.shape[-1]
    return torch.cat([y[:, :n0].reshape(B * n0, F), y[:, n0:].reshape(B * n1, F)], 0)


def reference(c, layer, mode_sep, qkv=None):
    B, n0, n1, H, D = c["B"], c["n0"], c["n1"], c["H"], c["D"]
    N = n0 + n1
    x = to_seq((c["qkv"] if qkv is None else qkv).float(), c).view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    q, k, v = x[0] * 0.125, x[1], x[2]
    s = q @ k.transpose(-1, -2)
    if c["table"] is not None:
        pos = torch.cat([torch.arange(n0, device="cuda"), c["pos1"] + torch.arange(n1, device="cuda")])
        ii = c["idx"].long()[pos][:, pos]
        bias = c["table"][:, layer * H:(layer + 1) * H][ii]  # [N,N,H]
        s = s + bias.permute(2, 0, 1)[None]
    keep = torch.ones(B, N, dtype=torch.bool, device="cuda")
    if c["keep0"] is not None:
        keep[:, :n0] = c["keep0"].bool()
    if c.get("keep1") is not None:
        keep[:, n0:] = c["keep1"].bool()
    s = s.masked_fill(~keep[:, None, None, :], float("-inf"))
    if mode_sep:
        blk = torch.zeros(N, N, dtype=torch.bool, device="cuda")
        blk[:n0, :n0] = True
        blk[n0:, n0:] = True
        s = s.masked_fill(~blk[None, None], float("-inf"))
    p = s.softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(B, N, D)
    return from_seq(o, c), s, p


CASES = [
    dict(B=2, n0=40, n1=197, H=3),
    dict(B=2, n0=0, n1=577, H=2),
    dict(B=3, n0=40, n1=0, H=3),
    dict(B=1, n0=12, n1=70, H=1),
    dict(B=2, n0=40, n1=577, H=12),
    dict(B=2, n0=40, n1=901, H=2),   # 480^2 (the reference's VQA geometry, README.md:194-223): 941 positions, 8 stationary tiles
]


@pytest.mark.parametrize("ci", range(len(CASES)))
@pytest.mark.parametrize("sep", [False, True])
@pytest.mark.parametrize("with_bias", [True, False])
def test_attention_fwd(ops, L, ci, sep, with_bias):
    c = build_case(seed=ci * 10 + sep, with_bias=with_bias, **CASES[ci])
    seq = ops.Seq(c["B"], c["n0"], c["n1"])
    assert seq.pos1 == c["pos1"]
    rows = seq.rows
    out = torch.full((rows, c["D"]), float("nan"), device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(c["H"], rows, device="cuda")
    layer = 1
    bias_t = c["table"].t().contiguous() if with_bias else None
    ops.attention_fwd(c["qkv"], out, lse, seq, c["H"], bias_t=bias_t, head_row0=layer * c["H"],
                      rel_index=c["idx"] * 4 if with_bias else None, rel_index_t=make_idx_t(c) if with_bias else None,
                      keep0=c["keep0"], mode=L.ATTN_SEPARATE if sep else L.ATTN_JOINT)
    torch.cuda.synchronize()
    ref, s, _ = reference(c, layer, sep)
    vmax = float(c["qkv"].float().abs().max())
    err = (out.float() - ref).abs()
    tol = 2e-2 * vmax + 1e-2 * ref.abs()
    assert not torch.isnan(out.float()).any()
    assert bool((err <= tol).all()), "max err %.4g" % float(err.max())
    # lse (log2 domain) against the reference's logsumexp
    ref_lse = from_seq(torch.logsumexp(s, -1).permute(0, 2, 1), c).t() * 1.4426950408889634
    # the kernel multiplies Q by scale*log2(e) before the MFMA (one more bf16 rounding of q: 2^-9 relative on scores of
    # magnitude <= ~20 here) and adds the bias as fp16(bias*log2 e)
    assert torch.allclose(lse, ref_lse, rtol=1e-3, atol=3e-2)


@pytest.mark.parametrize("ci", range(len(CASES)))
@pytest.mark.parametrize("sep", [False, True])
@pytest.mark.parametrize("with_bias", [True, False])
def test_attention_bwd(ops, L, ci, sep, with_bias):
    """Backward vs torch autograd of the fp32 restatement.  Tolerance: P, dS, dO enter the MFMAs as bf16 and the
    result is stored as bf16: 3e-2 of the tensor's max magnitude absolute + 3e-2 relative."""
    c = build_case(seed=100 + ci * 10 + sep, with_bias=with_bias, **CASES[ci])
    seq = ops.Seq(c["B"], c["n0"], c["n1"])
    rows, H, D = seq.rows, c["H"], c["D"]
    layer = 1
    g = torch.Generator(device="cuda"); g.manual_seed(7 + ci)
    dout = torch.randn(rows, D, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(H, rows, device="cuda")
    bias_t = c["table"].t().contiguous() if with_bias else None
    mode = L.ATTN_SEPARATE if sep else L.ATTN_JOINT
    kw = dict(bias_t=bias_t, head_row0=layer * H, rel_index=c["idx"] * 4 if with_bias else None,
              rel_index_t=make_idx_t(c) if with_bias else None, keep0=c["keep0"], mode=mode)
    ops.attention_fwd(c["qkv"], out, lse, seq, H, **kw)
    dqkv = torch.zeros(rows, 3 * D, device="cuda", dtype=torch.bfloat16)
    dbias_t = torch.zeros_like(bias_t) if with_bias else None
    # fused q_bias / v_bias gradients: per-segment column sums of dQ / dV, accumulated onto existing values
    csq = [torch.full((D,), 0.5, device="cuda"), torch.full((D,), -1.0, device="cuda")]
    csv = [torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")]
    ops.attention_bwd(c["qkv"], out, dout, lse, dqkv, seq, H, dbias_t=dbias_t, dq_colsum=csq, dv_colsum=csv, **kw)
    torch.cuda.synchronize()
    n_text = seq.B * seq.n0
    for sgm, rows_s in ((0, slice(0, n_text)), (1, slice(n_text, rows))):
        for name, got, sl, init in (("dq", csq[sgm], slice(0, D), (0.5, -1.0)[sgm]), ("dv", csv[sgm], slice(2 * D, 3 * D), 0.0)):
            want = dqkv[rows_s, sl].float().sum(0)  # the kernel sums the fp32 values it rounds to bf16 for dqkv
            err = (got - init - want).abs()
            lim = 2e-2 * float(want.abs().max()) + 0.05
            assert float(err.max()) <= lim, "%s colsum seg %d: err %.4g lim %.4g" % (name, sgm, float(err.max()), lim)
    # reference
    q32 = c["qkv"].float().requires_grad_(True)
    tab = c["table"].clone().requires_grad_(True) if with_bias else None
    cc = dict(c); cc["table"] = tab
    ref_o, _, _ = reference(cc, layer, sep, qkv=q32)
    (ref_o * dout.float()).sum().backward()
    ref = q32.grad
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        a, b = dqkv[:, sl].float(), ref[:, sl]
        err = (a - b).abs()
        tol = 3e-2 * float(b.abs().max()) + 3e-2 * b.abs()
        assert bool((err <= tol).all()), "%s max err %.4g (ref max %.4g)" % (name, float(err.max()), float(b.abs().max()))
    if with_bias:
        gt = tab.grad.t()  # [2H, R]
        a, b = dbias_t[layer * H:(layer + 1) * H], gt[layer * H:(layer + 1) * H]
        err = (a - b).abs()
        tol = 2e-2 * float(b.abs().max()) + 2e-2 * b.abs()
        assert bool((err <= tol).all()), "dbias max err %.4g (ref max %.4g)" % (float(err.max()), float(b.abs().max()))
        assert float(dbias_t[:H].abs().max()) == 0.0  # other layer's rows untouched


@pytest.mark.parametrize("ci", range(len(CASES)))
@pytest.mark.parametrize("sep", [False, True])
def test_bias_dense_tables(ops, L, ci, sep):
    """vlm_bias_dense: fp16 table[index] * log2(e) in the tiled MFMA-operand order documented in attention_common.h
    (one 4-KiB tile per 32 stationary x 64 streamed positions: [blk][j][lane][8]), both orientations; positions that
    are not members of the tile's segment (gap, past the end, SEPARATE: the other segment) hold a large negative value."""
    c = build_case(seed=300 + ci * 10, with_bias=True, **CASES[ci])
    bias_t = c["table"].t().contiguous()
    idx = (c["idx"] * 4).contiguous()
    n0, n1, pos1 = c["n0"], c["n1"], c["pos1"]
    NP = pos1 + n1
    seq = ops.Seq(c["B"], n0, n1)
    mode = L.ATTN_SEPARATE if sep else L.ATTN_JOINT
    d = ops.bias_dense(bias_t, idx, seq, mode)
    if sep:
        parts = [(0, n0, (n0 + 127) // 128 * 4, (n0 + 63) // 64), (pos1, NP, (n1 + 127) // 128 * 4, (n1 + 63) // 64)]
    else:
        parts = [(0, NP, (NP + 127) // 128 * 4, (NP + 63) // 64)]
    assert d.tiles == sum(nsb * nst for _, _, nsb, nst in parts)
    full = bias_t[:, (idx[:NP, :NP].long() >> 2)] * 1.4426950408889634   # [cols, q, k]
    lane = torch.arange(64, device="cuda")
    for got, k_major in ((d.q_major, False), (d.k_major, True)):
        t = got.view(bias_t.shape[0], d.tiles, 2, 2, 64, 8).float()
        first = 0
        for org, lim, nsb, nst in parts:
            for sb in range(nsb):
                for st in range(nst):
                    tile = t[:, first + sb * nst + st]                                  # [cols, blk, j, lane, e]
                    s_pos = org + 32 * sb + (lane & 31)                                   # [64]
                    blk, j, e = torch.meshgrid(torch.arange(2), torch.arange(2), torch.arange(8), indexing="ij")
                    t_pos = (org + 64 * st + 32 * blk[..., None] + 16 * (lane >> 5).cpu()[None, None, None, :]
                             + 8 * j[..., None] + e[..., None]).permute(0, 1, 3, 2).cuda()  # [blk, j, lane, e]
                    sp = s_pos[None, None, :, None].expand_as(t_pos)

                    def member(p_):
                        return (p_ >= org) & (p_ < lim) & ((p_ < n0) | (p_ >= pos1)) & (p_ < NP)
                    ok = member(sp) & member(t_pos)
                    q, k = (t_pos, sp) if k_major else (sp, t_pos)
                    want = torch.where(ok[None], full[:, q.clamp(0, NP - 1), k.clamp(0, NP - 1)], torch.full((), -30000.0, device="cuda"))
                    assert torch.allclose(tile, want, rtol=1e-3, atol=1e-3), (k_major, sb, st)
            first += nsb * nst


@pytest.mark.parametrize("k", [14, -24])
def test_bias_gradient_fixed_point_histogram_follows_the_magnitude(ops, L, k):
    """attn_bwd_dbias16_kernel sums dS in 64-bit fixed-point LDS bins whose step is 2^-48 of the work item's largest value:
    scaling dO by 2^k (exact in bf16 and in every product on the way) must scale the bias-table gradient by 2^k -- no
    overflow at large gradients, no underflow to zero at tiny ones (a fixed step would fail one of the two)."""
    c = build_case(seed=4242, **CASES[-1])
    seq = ops.Seq(c["B"], c["n0"], c["n1"])
    rows, H, D = seq.rows, c["H"], c["D"]
    g = torch.Generator(device="cuda"); g.manual_seed(99)
    dout = torch.randn(rows, D, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(H, rows, device="cuda")
    bias_t = c["table"].t().contiguous()
    kw = dict(bias_t=bias_t, head_row0=H, rel_index=c["idx"] * 4, rel_index_t=make_idx_t(c), keep0=c["keep0"], mode=L.ATTN_JOINT)
    ops.attention_fwd(c["qkv"], out, lse, seq, H, **kw)
    res = []
    for scale in (1.0, 2.0 ** k):
        dqkv = torch.zeros(rows, 3 * D, device="cuda", dtype=torch.bfloat16)
        dbias_t = torch.zeros_like(bias_t)
        ops.attention_bwd(c["qkv"], out, (dout.float() * scale).to(torch.bfloat16), lse, dqkv, seq, H, dbias_t=dbias_t, **kw)
        torch.cuda.synchronize()
        res.append(dbias_t[H:2 * H].double() / scale)
    ref, got = res
    assert float(ref.abs().max()) > 0
    # the items' sums reach the table through float atomics in arrival order: last-bit differences only
    assert float((got - ref).abs().max()) <= 1e-5 * float(ref.abs().max())  # (75 fp32 additions per bin, any order: <= 4.5e-6)


@pytest.mark.parametrize("sep", [False, True])
def test_attention_with_an_image_keep_mask(ops, L, sep):
    """keep1 (dropped IMAGE tokens: infer(image_embeds=, image_masks=), vilt_module.py:1092-1108) through the forward kernel
    and the three backward kernels, against the same fp32 restatement; the last image tokens of every sample and a few in
    the middle are dropped."""
    c = build_case(seed=777 + sep, **CASES[0])
    B, n0, n1, H, D = c["B"], c["n0"], c["n1"], c["H"], c["D"]
    keep1 = torch.ones(B, n1, dtype=torch.uint8, device="cuda")
    keep1[:, -37:] = 0
    keep1[0, 50:53] = 0
    c["keep1"] = keep1
    seq = ops.Seq(B, n0, n1)
    rows, layer = seq.rows, 1
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    dout = torch.randn(rows, D, device="cuda", generator=g).to(torch.bfloat16)
    out = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(H, rows, device="cuda")
    bias_t = c["table"].t().contiguous()
    kw = dict(bias_t=bias_t, head_row0=layer * H, rel_index=c["idx"] * 4, rel_index_t=make_idx_t(c), keep0=c["keep0"], keep1=keep1,
              mode=L.ATTN_SEPARATE if sep else L.ATTN_JOINT)
    ops.attention_fwd(c["qkv"], out, lse, seq, H, **kw)
    dqkv = torch.zeros(rows, 3 * D, device="cuda", dtype=torch.bfloat16)
    dbias_t = torch.zeros_like(bias_t)
    ops.attention_bwd(c["qkv"], out, dout, lse, dqkv, seq, H, dbias_t=dbias_t, **kw)
    torch.cuda.synchronize()
    q32 = c["qkv"].float().requires_grad_(True)
    tab = c["table"].clone().requires_grad_(True)
    cc = dict(c); cc["table"] = tab
    ref_o, _, _ = reference(cc, layer, sep, qkv=q32)
    vmax = float(c["qkv"].float().abs().max())
    assert bool(((out.float() - ref_o).abs() <= 2e-2 * vmax + 1e-2 * ref_o.abs()).all())
    (ref_o * dout.float()).sum().backward()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        a, b = dqkv[:, sl].float(), q32.grad[:, sl]
        assert bool(((a - b).abs() <= 3e-2 * float(b.abs().max()) + 3e-2 * b.abs()).all()), name
    # dropped image keys receive no gradient at all
    img = dqkv[B * n0:].view(B, n1, 3 * D)
    assert float(img[:, -37:, D:].abs().max()) == 0.0
    a, b = dbias_t[layer * H:(layer + 1) * H], tab.grad.t()[layer * H:(layer + 1) * H]
    assert bool(((a - b).abs() <= 2e-2 * float(b.abs().max()) + 2e-2 * b.abs()).all())


def test_attention_480_geometry_with_the_reference_index_and_ragged_masks(ops, L, pkg, golden_dir):
    """N = 941 (480^2, 30 x 30 patches) at the benchmark's batch of 22 with ragged text masks, on the REFERENCE's own index
    (index_buffers.npz `text_imag_relative_position_index_480`, R = 3 878 rows): forward, dQ / dK / dV and the bias-table gradient
    against the fp32 restatement -- the int16 offsets (4 R = 15 512), the histogram's LDS bound and the tiling at 8 stationary
    tiles are exercised, not only argued."""
    import os
    import numpy as np
    vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    z = np.load(os.path.join(golden_dir, "index_buffers.npz"))
    ref_idx = torch.from_numpy(z["text_imag_relative_position_index_480"]).long()
    B, n0, n1, H = 22, 40, 901, 2
    R = int(ref_idx.max()) + 1
    assert R == 59 * 59 + 3 + 392 + 2
    c = build_case(B, n0, n1, H, seed=480)
    m16, m16t = vm._index16(ref_idx.float(), n0)          # the model's own conversion (index coordinates, x 4)
    g = torch.Generator(device="cuda"); g.manual_seed(481)
    c["table"] = torch.randn(R, 2 * H, device="cuda", generator=g)
    c["R"] = R
    pos1 = c["pos1"]
    NP = pos1 + n1
    idx = torch.zeros(NP, (NP + 3) // 4 * 4, dtype=torch.int16, device="cuda")
    pos = torch.cat([torch.arange(n0), pos1 + torch.arange(n1)]).cuda()
    idx[pos[:, None], pos[None, :]] = ref_idx.to(torch.int16).cuda()
    c["idx"] = idx
    assert torch.equal(m16.cuda()[pos][:, pos].long(), 4 * ref_idx.cuda()) and torch.equal(m16t.cuda()[pos][:, pos].long(), 4 * ref_idx.cuda().t())
    seq = ops.Seq(B, n0, n1)
    rows, D, layer = seq.rows, c["D"], 1
    out = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
    lse = torch.empty(H, rows, device="cuda")
    bias_t = c["table"].t().contiguous()
    kw = dict(bias_t=bias_t, head_row0=layer * H, rel_index=m16.cuda().contiguous(), rel_index_t=m16t.cuda().contiguous(),
              keep0=c["keep0"], mode=L.ATTN_JOINT)
    ops.attention_fwd(c["qkv"], out, lse, seq, H, **kw)
    dout = torch.randn(rows, D, device="cuda", generator=g).to(torch.bfloat16)
    dqkv = torch.zeros(rows, 3 * D, device="cuda", dtype=torch.bfloat16)
    dbias_t = torch.zeros_like(bias_t)
    ops.attention_bwd(c["qkv"], out, dout, lse, dqkv, seq, H, dbias_t=dbias_t, **kw)
    torch.cuda.synchronize()
    q32 = c["qkv"].float().requires_grad_(True)
    tab = c["table"].clone().requires_grad_(True)
    cc = dict(c); cc["table"] = tab
    ref_o, _, _ = reference(cc, layer, False, qkv=q32)
    vmax = float(c["qkv"].float().abs().max())
    err = (out.float() - ref_o.detach()).abs()
    assert bool((err <= 2e-2 * vmax + 1e-2 * ref_o.detach().abs()).all()), "forward max err %.4g" % float(err.max())
    (ref_o * dout.float()).sum().backward()
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        a, b = dqkv[:, sl].float(), q32.grad[:, sl]
        e = (a - b).abs()
        assert bool((e <= 3e-2 * float(b.abs().max()) + 3e-2 * b.abs()).all()), "%s max err %.4g" % (name, float(e.max()))
    a, b = dbias_t[layer * H:(layer + 1) * H], tab.grad.t()[layer * H:(layer + 1) * H]
    e = (a - b).abs()
    assert bool((e <= 2e-2 * float(b.abs().max()) + 2e-2 * b.abs()).all()), "dbias max err %.4g (ref max %.4g)" % (
        float(e.max()), float(b.abs().max()))
    assert float(dbias_t[:H].abs().max()) == 0.0
