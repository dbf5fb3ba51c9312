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
    pos1 = (n0 + 3) // 4 * 4
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
    assert torch.allclose(lse, ref_lse, rtol=1e-4, atol=2e-3)


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
def test_attention_dense_bias_matches_gather(ops, L, ci, sep):
    """Dense fp16 bias mode (vlm_bias_dense + BIAS == 2 kernels) against the LDS-gather mode on the same inputs: same
    scores up to the fp16 rounding of the bias (2^-11 relative), i.e. well inside the bf16 tolerance of the outputs;
    and the dense matrix itself against table[index] * log2(e)."""
    c = build_case(seed=300 + ci * 10 + sep, with_bias=True, **CASES[ci])
    seq = ops.Seq(c["B"], c["n0"], c["n1"])
    rows, H, D = seq.rows, c["H"], c["D"]
    layer = 1
    bias_t = c["table"].t().contiguous()
    idx, idx_t = (c["idx"] * 4).contiguous(), make_idx_t(c).contiguous()
    dense = (ops.bias_dense(bias_t, idx), ops.bias_dense(bias_t, idx_t))
    want = bias_t[:, (idx.long() >> 2).clamp_(0, bias_t.shape[1] - 1)] * 1.4426950408889634
    assert torch.allclose(dense[0].float(), want, rtol=1e-3, atol=1e-4)
    mode = L.ATTN_SEPARATE if sep else L.ATTN_JOINT
    g = torch.Generator(device="cuda"); g.manual_seed(11 + ci)
    dout = torch.randn(rows, D, device="cuda", generator=g).to(torch.bfloat16)
    res = []
    for bd in (None, dense):
        kw = dict(bias_t=bias_t, head_row0=layer * H, rel_index=idx, rel_index_t=idx_t, keep0=c["keep0"], mode=mode,
                  bias_dense=bd)
        out = torch.empty(rows, D, device="cuda", dtype=torch.bfloat16)
        lse = torch.empty(H, rows, device="cuda")
        ops.attention_fwd(c["qkv"], out, lse, seq, H, **kw)
        dqkv = torch.zeros(rows, 3 * D, device="cuda", dtype=torch.bfloat16)
        dbias_t = torch.zeros_like(bias_t)
        ops.attention_bwd(c["qkv"], out, dout, lse, dqkv, seq, H, dbias_t=dbias_t, **kw)
        res.append((out.float(), lse, dqkv.float(), dbias_t))
    (o0, l0, d0, b0), (o1, l1, d1, b1) = res
    assert float((o0 - o1).abs().max()) <= 2e-2 * float(o0.abs().max()) + 1e-3
    assert torch.allclose(l0, l1, rtol=1e-3, atol=5e-3)
    assert float((d0 - d1).abs().max()) <= 2e-2 * float(d0.abs().max()) + 1e-3
    assert float((b0 - b1).abs().max()) <= 2e-2 * float(b0.abs().max()) + 1e-3
