"""Thin typed wrappers over the C ABI (one Python function per extern "C" entry point).

Every wrapper takes torch CUDA tensors, checks dtype/contiguity, and launches on torch's current stream.
No wrapper has a CPU path.
"""
import ctypes

import torch

from . import _lib as L

BF16 = torch.bfloat16
F32 = torch.float32


def _ld(t):
    """Leading dimension (row stride in elements) of a 2-D row-major view."""
    if t.dim() != 2 or t.stride(1) != 1:
        raise L.VlmError("expected a 2-D tensor with unit inner stride, got shape %s strides %s"
                         % (tuple(t.shape), t.stride()))
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def gemm(a, b, out, ta=False, tb=False, *, bias=None, act=L.ACT_NONE, aux=None, col_scale=None, row_scale=None,
         residual=None, alpha=1.0, accumulate=False, col_sum=None, col_sum_fold=None):
    """out[M,N] = epilogue(op(a)[M,K] @ op(b)[K,N]); see include/vlm_hip.h for the epilogue algebra.

    a: bf16 [M,K] (ta=False) or [K,M] (ta=True);  b: bf16 [N,K] (tb=False, nn.Linear layout) or [K,N] (tb=True).
    """
    L.require_cuda(a, b, out, bias, aux, col_scale, row_scale, residual, col_sum)
    if a.dtype != BF16 or b.dtype != BF16:
        raise L.VlmError("gemm operands must be bfloat16")
    M, N = out.shape
    K = a.shape[0] if ta else a.shape[1]
    if (a.shape[1] if ta else a.shape[0]) != M:
        raise L.VlmError("gemm: A rows %s do not match M=%d" % (tuple(a.shape), M))
    kb, nb = (b.shape[0], b.shape[1]) if tb else (b.shape[1], b.shape[0])
    if nb != N or kb != K:
        raise L.VlmError("gemm: B shape %s incompatible with N=%d K=%d" % (tuple(b.shape), N, K))
    e = L.Epilogue()
    e.bias = bias.data_ptr() if bias is not None else 0
    e.col_scale = col_scale.data_ptr() if col_scale is not None else 0
    e.row_scale = row_scale.data_ptr() if row_scale is not None else 0
    e.residual = residual.data_ptr() if residual is not None else 0
    e.ld_res = _ld(residual) if residual is not None else 0
    e.aux = aux.data_ptr() if aux is not None else 0
    e.ld_aux = _ld(aux) if aux is not None else 0
    e.act = act
    e.alpha = alpha
    e.accumulate = 1 if accumulate else 0
    e.col_sum = col_sum.data_ptr() if col_sum is not None else 0
    region = None
    if col_sum is not None and col_sum_fold is not None and not ta and N % 128 == 0 and M >= 128:
        # complete 128-row tiles park their column sums in the fold batch's scratch (plain stores), one
        # vlm_colreduce_batch later adds them to col_sum together with the block's other column partials
        region = col_sum_fold.next_region()
        if region.numel() >= (M // 128) * 2 * N:
            e.col_sum_ws = region.data_ptr()
        else:
            region = None
    if ta and tb and out.dtype == F32 and K >= 2048 and bias is None and col_sum is None:
        ws = _splitk_workspace(out.device)
        e.splitk_ws = ws.data_ptr()
        e.splitk_ws_bytes = ws.numel() * 4
    if col_sum is not None and (col_sum.numel() < N or not col_sum.is_contiguous()):
        raise L.VlmError("gemm: col_sum must be a contiguous f32 vector of at least N elements")
    for t, dt in ((bias, F32), (col_scale, F32), (row_scale, F32), (residual, F32), (aux, BF16), (col_sum, F32)):
        if t is not None and t.dtype != dt:
            raise L.VlmError("gemm epilogue tensor has dtype %s, expected %s" % (t.dtype, dt))
    if out.dtype not in (BF16, F32):
        raise L.VlmError("gemm output must be bf16 or f32")
    rc = L.get_lib().vlm_gemm_bf16(int(ta), int(tb), M, N, K, L.ptr(a), _ld(a), L.ptr(b), _ld(b), L.ptr(out), _ld(out),
                                   int(out.dtype == F32), ctypes.byref(e), L.stream_ptr())
    L.check(rc, "vlm_gemm_bf16")
    if region is not None:
        col_sum_fold.add(region, M // 128, N, col_sum, None)
    return out


def gemm_grouped(a, groups, out, *, act=L.ACT_NONE, aux=None, col_scale=None, row_scale=None, residual=None, alpha=1.0,
                 col_sum_fold=None):
    """out[r0:r1] = epilogue(a[r0:r1] @ w_g^T) for every group (r0, r1, w_g [N, K] bf16, bias_g or None, col_sum_g or None) in
    ONE launch (include/vlm_hip.h vlm_gemm_bf16_grouped): the modality experts of an all_moe block
    (vision_transformer.py:607-681).  aux / col_scale / row_scale / residual are whole-matrix epilogue inputs as in gemm()."""
    L.require_cuda(a, out, aux, col_scale, row_scale, residual)
    if a.dtype != BF16:
        raise L.VlmError("gemm operands must be bfloat16")
    if len(groups) < 1 or len(groups) > L.GEMM_MAX_GROUPS:
        raise L.VlmError("gemm_grouped: 1..%d groups" % L.GEMM_MAX_GROUPS)
    M, N = out.shape
    K = a.shape[1]
    if a.shape[0] != M:
        raise L.VlmError("gemm_grouped: A rows %s do not match M=%d" % (tuple(a.shape), M))
    e = L.Epilogue()
    e.col_scale = col_scale.data_ptr() if col_scale is not None else 0
    e.row_scale = row_scale.data_ptr() if row_scale is not None else 0
    e.residual = residual.data_ptr() if residual is not None else 0
    e.ld_res = _ld(residual) if residual is not None else 0
    e.aux = aux.data_ptr() if aux is not None else 0
    e.ld_aux = _ld(aux) if aux is not None else 0
    e.act = act
    e.alpha = alpha
    for t, dt in ((col_scale, F32), (row_scale, F32), (residual, F32), (aux, BF16)):
        if t is not None and t.dtype != dt:
            raise L.VlmError("gemm epilogue tensor has dtype %s, expected %s" % (t.dtype, dt))
    if out.dtype not in (BF16, F32):
        raise L.VlmError("gemm output must be bf16 or f32")
    arr = (L.GemmGroup * len(groups))()
    folds = []
    prev = 0
    if col_sum_fold is not None and len(col_sum_fold.jobs) + len(groups) > col_sum_fold.MAX:
        col_sum_fold.flush()  # never in the middle of this call: its regions are reserved before the launch that fills them
    for g, (r0, r1, w, bias, col_sum) in zip(arr, groups):
        L.require_cuda(w, bias, col_sum)
        if w.dtype != BF16 or tuple(w.shape) != (N, K) or r0 < prev or r1 < r0 or r1 > M:
            raise L.VlmError("gemm_grouped: group (%d, %d) weight %s for N=%d K=%d M=%d" % (r0, r1, tuple(w.shape), N, K, M))
        prev = r1
        for t in (bias, col_sum):
            if t is not None and (t.dtype != F32 or t.numel() < N or not t.is_contiguous()):
                raise L.VlmError("gemm_grouped: bias / col_sum must be contiguous f32 vectors of at least N elements")
        g.row0, g.rows, g.B, g.ldb = r0, r1 - r0, w.data_ptr(), _ld(w)
        g.bias = bias.data_ptr() if bias is not None else None
        g.col_sum = col_sum.data_ptr() if col_sum is not None else None
        rows = r1 - r0
        if col_sum is not None and col_sum_fold is not None and N % 128 == 0 and rows >= 128:
            region = col_sum_fold.next_region()
            if region.numel() >= (rows // 128) * 2 * N:
                g.col_sum_ws = region.data_ptr()
                # reserve the region NOW (the next group's next_region() must not hand out the same one)
                col_sum_fold.add(region, rows // 128, N, col_sum, None)
                folds.append(region)
    rc = L.get_lib().vlm_gemm_bf16_grouped(len(groups), arr, N, K, L.ptr(a), _ld(a), L.ptr(out), _ld(out),
                                           int(out.dtype == F32), ctypes.byref(e), L.stream_ptr())
    L.check(rc, "vlm_gemm_bf16_grouped")
    return out


def gemm_wgrad_grouped(dy, x, groups, accumulate=True):
    """dW_g (+)= dy[r0:r1]^T x[r0:r1] for every group (r0, r1, dW_g f32 [M, N]) of token rows in one launch
    (include/vlm_hip.h vlm_gemm_wgrad_grouped): the weight gradients of an all_moe block's experts."""
    L.require_cuda(dy, x)
    if dy.dtype != BF16 or x.dtype != BF16 or dy.shape[0] != x.shape[0]:
        raise L.VlmError("gemm_wgrad_grouped: bf16 dy [tokens, M] and x [tokens, N]")
    M, N = dy.shape[1], x.shape[1]
    arr = (L.WgradGroup * len(groups))()
    prev, ldc = 0, None
    for g, (r0, r1, dW) in zip(arr, groups):
        L.require_cuda(dW)
        if dW.dtype != F32 or tuple(dW.shape) != (M, N) or r0 < prev or r1 < r0 or r1 > dy.shape[0]:
            raise L.VlmError("gemm_wgrad_grouped: group (%d, %d) gradient %s for M=%d N=%d" % (r0, r1, tuple(dW.shape), M, N))
        if ldc is not None and _ld(dW) != ldc:
            raise L.VlmError("gemm_wgrad_grouped: the groups' gradients must share one leading dimension")
        ldc = _ld(dW)
        prev = r1
        g.row0, g.rows, g.C, g.accumulate = r0, r1 - r0, dW.data_ptr(), int(bool(accumulate))
    ws = _splitk_workspace(dy.device)
    rc = L.get_lib().vlm_gemm_wgrad_grouped(len(groups), arr, M, N, L.ptr(dy), _ld(dy), L.ptr(x), _ld(x), ldc, L.ptr(ws),
                                            ws.numel() * 4, L.stream_ptr())
    L.check(rc, "vlm_gemm_wgrad_grouped")


_SPLITK_WS = {}
SPLITK_WS_BYTES = 96 << 20


def _splitk_workspace(device):
    """Per (device, stream) scratch for the wgrad GEMM's K slices (vlm_epilogue_t.splitk_ws): fp32 [slice][M][N] tiles,
    at most 256 CUs / tiles slices of an M x N <= 3072 x 768 weight -- 66 MB at the base width."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _SPLITK_WS.get(key)
    if ws is None:
        ws = _SPLITK_WS[key] = torch.empty(SPLITK_WS_BYTES // 4, device=device, dtype=F32)
    return ws


def layernorm_fwd(x, gamma, beta, eps, out, stats=None):
    L.require_cuda(x, gamma, beta, out, stats)
    M, D = x.shape
    rc = L.get_lib().vlm_layernorm_fwd(L.ptr(x), _ld(x), M, D, L.ptr(gamma), L.ptr(beta), eps, L.ptr(out), _ld(out),
                                       int(out.dtype == F32), L.ptr(stats), L.stream_ptr())
    L.check(rc, "vlm_layernorm_fwd")
    return out


_ROW_WS = {}


def _row_workspace(device, D, slot=0):
    """Per-device scratch for the row kernels' per-workgroup column partials (VLM_ROW_WS_BYTES); slot 1: the second
    partial region of a fused call."""
    key = (device.index, torch.cuda.current_stream().cuda_stream, slot)
    n = 1536 * 2 * max(D, 1024)
    ws = _ROW_WS.get(key)
    if ws is None or ws.numel() < n:
        ws = _ROW_WS[key] = torch.empty(n, device=device, dtype=F32)
    return ws


class FoldBatch:
    """Deferred column-sum folds of several row kernels (vlm_colreduce_batch): each deferred call writes its
    per-workgroup partials to its own region of one scratch tensor; flush() folds them all in ONE launch."""

    MAX = 16  # VLM_MAX_FOLD_JOBS: an all_moe block's backward parks 10 (two experts x five row kernels / epilogues)

    def __init__(self, device, D=1024):
        self.region = 1536 * 2 * max(D, 1024)
        key = ("fold", device.index, torch.cuda.current_stream().cuda_stream)
        ws = _ROW_WS.get(key)
        if ws is None or ws.numel() < self.MAX * self.region:
            ws = _ROW_WS[key] = torch.empty(self.MAX * self.region, device=device, dtype=F32)
        self.ws = ws
        self.jobs = []
        self.taken = 0  # regions handed out since the last flush (a region whose job turned out empty stays taken)
        self.multi_stream = False  # set by a caller whose row kernels run on more than one stream between two flushes

    def next_region(self):
        return self.next_regions(1)[0]

    def next_regions(self, n):
        """n consecutive free regions (a fused row kernel parks one job per column-sum pair): all reserved BEFORE the launch,
        because making room (flush) after the first was taken would fold a region the launch has not written yet."""
        if self.taken + n > self.MAX:
            if self.multi_stream:
                raise L.VlmError("FoldBatch overflow while its producers run on several streams (a mid-way fold would "
                                 "read partials of launches the folding stream is not ordered behind)")
            self.flush()
        i = self.taken
        self.taken += n
        return [self.ws[(i + k) * self.region:(i + k + 1) * self.region] for k in range(n)]

    def add(self, region, nblocks, D, out0, out1):
        if nblocks > 0:
            self.jobs.append((region, nblocks, D, out0, out1))

    def flush(self):
        self.taken = 0
        if not self.jobs:
            return
        arr = (L.FoldJob * len(self.jobs))()
        for a, (region, nblocks, D, out0, out1) in zip(arr, self.jobs):
            a.partials, a.nblocks, a.D = region.data_ptr(), nblocks, D
            a.out0 = out0.data_ptr() if out0 is not None else None
            a.out1 = out1.data_ptr() if out1 is not None else None
        L.check(L.get_lib().vlm_colreduce_batch(arr, len(self.jobs), L.stream_ptr()), "vlm_colreduce_batch")
        self.jobs = []


def layernorm_bwd(dy, x, stats, gamma, dx, dres=None, dgamma=None, dbeta=None, fold=None):
    L.require_cuda(dy, x, stats, gamma, dx, dres, dgamma, dbeta)
    M, D = x.shape
    ws = fold.next_region() if fold is not None else _row_workspace(x.device, D)
    nb = ctypes.c_int(0)
    rc = L.get_lib().vlm_layernorm_bwd(L.ptr(dy), _ld(dy), int(dy.dtype == F32), L.ptr(x), _ld(x), L.ptr(stats),
                                       L.ptr(gamma), M, D, L.ptr(dres), _ld(dres) if dres is not None else 0,
                                       L.ptr(dx), _ld(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(ws), ws.numel() * 4,
                                       ctypes.byref(nb) if fold is not None else None, L.stream_ptr())
    L.check(rc, "vlm_layernorm_bwd")
    if fold is not None:
        fold.add(ws, nb.value, D, dgamma, dbeta)
    return dx


def layernorm_bwd_scale(dy, x, stats, gamma, dx, dres, dgamma, dbeta, *, y, sgamma, row_scale, sdy, dsgamma=None, dsbias=None,
                        fold=None):
    """layernorm_bwd(...) followed by layerscale_bwd(dx, y, sgamma, row_scale, sdy, dsgamma, dsbias) in ONE pass over the rows
    (vlm_layernorm_bwd_scale): the LayerScale backward of the branch below this LayerNorm reads the row while it is in
    registers.  Bit-identical to the two calls."""
    L.require_cuda(dy, x, stats, gamma, dx, dres, dgamma, dbeta, y, sgamma, row_scale, sdy, dsgamma, dsbias)
    M, D = x.shape
    nb = ctypes.c_int(0)
    if fold is not None:
        ws, ws2 = fold.next_regions(2)  # two jobs of the batch: this LayerNorm's sums and the LayerScale's
    else:
        ws, ws2 = _row_workspace(x.device, D), _row_workspace(x.device, D, slot=1)
    sc = L.LayerScale()
    sc.y, sc.ldy, sc.gamma, sc.row_scale = L.ptr(y), _ld(y) if y is not None else 0, L.ptr(sgamma), L.ptr(row_scale)
    sc.dy, sc.lddy, sc.dgamma, sc.dbias = L.ptr(sdy), _ld(sdy), L.ptr(dsgamma), L.ptr(dsbias)
    sc.workspace, sc.workspace_bytes = L.ptr(ws2), ws2.numel() * 4
    rc = L.get_lib().vlm_layernorm_bwd_scale(L.ptr(dy), _ld(dy), int(dy.dtype == F32), L.ptr(x), _ld(x), L.ptr(stats),
                                             L.ptr(gamma), M, D, L.ptr(dres), _ld(dres) if dres is not None else 0,
                                             L.ptr(dx), _ld(dx), L.ptr(dgamma), L.ptr(dbeta), L.ptr(ws), ws.numel() * 4,
                                             ctypes.byref(sc), ctypes.byref(nb) if fold is not None else None, L.stream_ptr())
    L.check(rc, "vlm_layernorm_bwd_scale")
    if fold is not None:
        # nb = the grid whenever EITHER partial set was written: a frozen LayerNorm (no dgamma / dbeta) still leaves the
        # LayerScale's partials in ws2
        fold.add(ws, nb.value if (dgamma is not None or dbeta is not None) else 0, D, dgamma, dbeta)
        fold.add(ws2, nb.value if (dsgamma is not None or dsbias is not None) else 0, D, dsgamma, dsbias)
    return dx


def layerscale_bwd(dx, y, gamma, row_scale, dy, dgamma=None, dbias=None, fold=None):
    """y None (then gamma / dgamma None too): the LayerScale is folded into the branch's projection (layerscale_fold): only
    dy = bf16(row_scale * dx) and its column sums (the raw bias gradient) remain."""
    L.require_cuda(dx, y, gamma, row_scale, dy, dgamma, dbias)
    M, D = dx.shape
    ws = fold.next_region() if fold is not None else _row_workspace(dx.device, D)
    nb = ctypes.c_int(0)
    rc = L.get_lib().vlm_layerscale_bwd(L.ptr(dx), _ld(dx), L.ptr(y), _ld(y) if y is not None else 0, L.ptr(gamma), L.ptr(row_scale), M, D,
                                        L.ptr(dy), _ld(dy), L.ptr(dgamma), L.ptr(dbias), L.ptr(ws), ws.numel() * 4,
                                        ctypes.byref(nb) if fold is not None else None, L.stream_ptr())
    L.check(rc, "vlm_layerscale_bwd")
    if fold is not None:
        fold.add(ws, nb.value, D, dgamma, dbias)
    return dy


def embedding_bwd(gy, ids, dW, padding_idx=-1):
    """dW[ids[t]] += gy[t] (fp32) for ids[t] != padding_idx: backward of an nn.Embedding gather into the flat gradient."""
    L.require_cuda(gy, ids, dW)
    n, D = gy.shape
    assert gy.dtype == F32 and dW.dtype == F32 and ids.dtype == torch.int64 and ids.numel() == n and gy.stride(1) == 1
    L.check(L.get_lib().vlm_embedding_bwd(L.ptr(gy), _ld(gy), L.ptr(ids), n, D, int(padding_idx), L.ptr(dW), _ld(dW),
                                          dW.shape[0], L.stream_ptr()), "vlm_embedding_bwd")
    return dW


def cross_entropy_fwd(logits, labels, ignore_index=-100):
    """(loss_rows f32 [rows], lse f32 [rows]) of bf16 logits [rows, V] (row stride a multiple of 8), include/vlm_hip.h."""
    L.require_cuda(logits, labels)
    rows, V = logits.shape
    if logits.dtype != BF16 or labels.dtype != torch.int64 or labels.numel() != rows or not labels.is_contiguous():
        raise L.VlmError("cross_entropy: bf16 logits [rows, V], contiguous int64 labels [rows]")
    loss_rows = torch.empty(rows, device=logits.device, dtype=F32)
    lse = torch.empty(rows, device=logits.device, dtype=F32)
    L.check(L.get_lib().vlm_cross_entropy_fwd(L.ptr(logits), _ld(logits), rows, V, L.ptr(labels), int(ignore_index), L.ptr(loss_rows),
                                              L.ptr(lse), L.stream_ptr()), "vlm_cross_entropy_fwd")
    return loss_rows, lse


def cross_entropy_bwd(logits, labels, lse, scale, ignore_index=-100):
    """dlogits (bf16 view [rows, V] of a zero-padded [rows, roundup(V, 64)] buffer) = scale[0] * (softmax - onehot), 0 in ignored
    rows; `scale` is a device scalar."""
    L.require_cuda(logits, labels, lse, scale)
    rows, V = logits.shape
    Vp = (V + 63) // 64 * 64
    buf = torch.empty(rows, Vp, device=logits.device, dtype=BF16)
    L.check(L.get_lib().vlm_cross_entropy_bwd(L.ptr(logits), _ld(logits), rows, V, L.ptr(labels), int(ignore_index), L.ptr(lse),
                                              L.ptr(scale), L.ptr(buf), Vp, L.stream_ptr()), "vlm_cross_entropy_bwd")
    return buf[:, :V]


def l2norm_fwd(x):
    """(y f32 [rows, D] = x / ||x||_2 per row, inv_norm f32 [rows]) of a bf16 / fp32 matrix with unit column stride."""
    L.require_cuda(x)
    rows, D = x.shape
    if x.dtype not in (BF16, F32) or x.stride(1) != 1:
        raise L.VlmError("l2norm: bf16 / fp32 [rows, D] with unit column stride")
    y = torch.empty(rows, D, device=x.device, dtype=F32)
    inv = torch.empty(rows, device=x.device, dtype=F32)
    L.check(L.get_lib().vlm_l2norm_fwd(L.ptr(x), int(x.dtype == BF16), _ld(x), rows, D, L.ptr(y), L.ptr(inv), L.stream_ptr()), "vlm_l2norm_fwd")
    return y, inv


def l2norm_bwd(g, y, inv, dtype):
    """dx = (g - y (g . y)) * inv_norm in `dtype` (bf16 / fp32), g and y fp32 contiguous."""
    L.require_cuda(g, y, inv)
    rows, D = y.shape
    if g.dtype != F32 or not g.is_contiguous() or g.shape != y.shape:
        raise L.VlmError("l2norm_bwd: contiguous fp32 gradient of y's shape")
    dx = torch.empty(rows, D, device=y.device, dtype=dtype)
    L.check(L.get_lib().vlm_l2norm_bwd(L.ptr(g), L.ptr(y), L.ptr(inv), rows, D, L.ptr(dx), int(dtype == BF16), D, L.stream_ptr()), "vlm_l2norm_bwd")
    return dx


def contrastive(all_img, all_txt, B, log_scale):
    """Symmetric contrastive loss of normalised features (include/vlm_hip.h vlm_contrastive): returns (out3 = [loss, d loss / d
    log_scale, exp(log_scale)], logits [n, n], d_img [B, D], d_txt [B, D])."""
    L.require_cuda(all_img, all_txt, log_scale)
    n, D = all_img.shape
    for t in (all_img, all_txt):
        if t.dtype != F32 or not t.is_contiguous() or t.shape != (n, D):
            raise L.VlmError("contrastive: contiguous fp32 [n, D] features")
    if log_scale.dtype != F32 or log_scale.numel() != 1:
        raise L.VlmError("contrastive: log_scale is one fp32 number")
    dev = all_img.device
    logits = torch.empty(n, n, device=dev, dtype=F32)
    out3 = torch.empty(3, device=dev, dtype=F32)
    d_img = torch.empty(B, D, device=dev, dtype=F32)
    d_txt = torch.empty(B, D, device=dev, dtype=F32)
    ws = torch.empty(max(1, L.get_lib().vlm_contrastive_ws_floats(n)), device=dev, dtype=F32)
    L.check(L.get_lib().vlm_contrastive(L.ptr(all_img), L.ptr(all_txt), n, B, D, L.ptr(log_scale), L.ptr(logits), L.ptr(out3), L.ptr(d_img),
                                        L.ptr(d_txt), L.ptr(ws), L.stream_ptr()), "vlm_contrastive")
    return out3, logits, d_img, d_txt


def small_cross_entropy(logits, labels):
    """(loss f32 [1], dlogits f32 [rows, V]) of F.cross_entropy(logits, labels) (mean) for small V; logits bf16 / fp32, unit column stride."""
    L.require_cuda(logits, labels)
    rows, V = logits.shape
    if logits.dtype not in (BF16, F32) or logits.stride(1) != 1 or labels.dtype != torch.int64 or not labels.is_contiguous() or labels.numel() != rows:
        raise L.VlmError("small_cross_entropy: bf16 / fp32 logits [rows, V], contiguous int64 labels [rows]")
    loss = torch.empty(1, device=logits.device, dtype=F32)
    d = torch.empty(rows, V, device=logits.device, dtype=F32)
    L.check(L.get_lib().vlm_small_cross_entropy(L.ptr(logits), int(logits.dtype == BF16), _ld(logits), rows, V, L.ptr(labels), L.ptr(loss),
                                                L.ptr(d), L.stream_ptr()), "vlm_small_cross_entropy")
    return loss, d


def cross_entropy_reduce(loss_rows, labels, V, ignore_index=-100):
    """out2 = [mean loss over the counted rows, 1 / count] (device)."""
    L.require_cuda(loss_rows, labels)
    out2 = torch.empty(2, device=loss_rows.device, dtype=F32)
    L.check(L.get_lib().vlm_cross_entropy_reduce(L.ptr(loss_rows), L.ptr(labels), loss_rows.numel(), int(V), int(ignore_index), L.ptr(out2),
                                                 L.stream_ptr()), "vlm_cross_entropy_reduce")
    return out2


def scale_by_scalar(tensors, scalar):
    """[t * scalar for t in tensors] (fp32 contiguous, at most four) in ONE launch; `scalar` is a device tensor with one element."""
    L.require_cuda(scalar, *tensors)
    k = len(tensors)
    if k == 0:
        return []
    if k > 4 or any(t.dtype != F32 or not t.is_contiguous() for t in tensors) or scalar.dtype != F32:
        raise L.VlmError("scale_by_scalar: up to four contiguous fp32 tensors and an fp32 device scalar")
    outs = [torch.empty_like(t) for t in tensors]
    P = ctypes.c_void_p * k
    I = ctypes.c_int * k
    L.check(L.get_lib().vlm_scale_by_scalar(P(*[L.ptr(t) for t in tensors]), P(*[L.ptr(o) for o in outs]), I(*[t.numel() for t in tensors]), k,
                                            L.ptr(scalar), L.stream_ptr()), "vlm_scale_by_scalar")
    return outs


_FRONT_WS = {}


def _front_ws(kind, D, device):
    """Per-(kind, D, device) scratch of the front-end backward kernels (column partials, folded in the same call)."""
    key = (kind, D, str(device))
    ws = _FRONT_WS.get(key)
    if ws is None:
        lib = L.get_lib()
        n = lib.vlm_text_rows_bwd_ws_floats(D) if kind == "text" else lib.vlm_image_rows_bwd_ws_floats(D)
        ws = _FRONT_WS[key] = torch.empty(max(1, n), device=device, dtype=F32)
    return ws


def _rowvec(t, D, what):
    if t is None:
        return 0
    if t.dtype != F32 or t.numel() != D or not t.is_contiguous():
        raise L.VlmError("%s: a contiguous fp32 vector of %d elements" % (what, D))
    return L.ptr(t)


def text_rows_fwd(ids, word, add0, gamma, beta, eps, out, u=None, p=0.0, scale=1.0, add1=None):
    """out[r] = dropout(LayerNorm(word[ids[r]] + add0)) + add1 (include/vlm_hip.h vlm_text_rows_fwd); out: fp32 [n, D] rows with unit
    column stride (a slice of the pass's token matrix); returns stats fp32 [n, 2]."""
    L.require_cuda(ids, word, add0, gamma, beta, out, u, add1)
    n, D = out.shape
    if ids.dtype != torch.int64 or ids.numel() != n or not ids.is_contiguous():
        raise L.VlmError("text_rows: contiguous int64 ids, one per output row")
    if word.dtype != F32 or word.shape[1] != D or word.stride(1) != 1 or out.dtype != F32 or out.stride(1) != 1:
        raise L.VlmError("text_rows: fp32 word table [V, D] and fp32 output rows")
    if u is not None and (u.dtype != F32 or u.numel() != n * D or not u.is_contiguous()):
        raise L.VlmError("text_rows: u is a contiguous fp32 [n, D]")
    stats = torch.empty(n, 2, device=out.device, dtype=F32)
    L.check(L.get_lib().vlm_text_rows_fwd(L.ptr(ids), n, L.ptr(word), _ld(word), _rowvec(add0, D, "add0"), _rowvec(gamma, D, "gamma"),
                                          _rowvec(beta, D, "beta"), float(eps), L.ptr(u) if u is not None else 0, float(p), float(scale),
                                          _rowvec(add1, D, "add1"), L.ptr(out), _ld(out), L.ptr(stats), D, L.stream_ptr()), "vlm_text_rows_fwd")
    return stats


def text_rows_bwd(g, ids, word, add0, gamma, stats, u, p, scale, d_word, padding_idx, d_add1, d_beta, d_gamma, d_add0):
    """Backward of text_rows_fwd: ADDS into d_word rows and the four vectors (None = not wanted)."""
    L.require_cuda(g, ids, word, add0, gamma, stats, u, d_word, d_add1, d_beta, d_gamma, d_add0)
    n, D = g.shape
    if g.dtype != F32 or g.stride(1) != 1:
        raise L.VlmError("text_rows_bwd: fp32 gradient rows with unit column stride")
    if d_word is not None and (d_word.dtype != F32 or d_word.shape != word.shape or d_word.stride() != word.stride()):
        raise L.VlmError("text_rows_bwd: d_word has the word table's layout")
    L.check(L.get_lib().vlm_text_rows_bwd(L.ptr(g), _ld(g), L.ptr(ids), n, L.ptr(word), _ld(word), _rowvec(add0, D, "add0"),
                                          _rowvec(gamma, D, "gamma"), L.ptr(stats), L.ptr(u) if u is not None else 0, float(p), float(scale), D,
                                          L.ptr(d_word) if d_word is not None else 0, -1 if padding_idx is None else int(padding_idx),
                                          _rowvec(d_add1, D, "d_add1"), _rowvec(d_beta, D, "d_beta"), _rowvec(d_gamma, D, "d_gamma"),
                                          _rowvec(d_add0, D, "d_add0"), L.ptr(_front_ws("text", D, g.device)), L.stream_ptr()),
            "vlm_text_rows_bwd")


def image_rows_prep(conv_bias, type_row, cls):
    """fp32 [2, D]: row 0 = conv_bias + type_row (the patch-embed GEMM's bias), row 1 = cls + type_row (the lead row)."""
    L.require_cuda(conv_bias, type_row, cls)
    D = cls.numel()
    out2 = torch.empty(2, D, device=cls.device, dtype=F32)
    L.check(L.get_lib().vlm_image_rows_prep(_rowvec(conv_bias, D, "conv_bias"), _rowvec(type_row, D, "type_row"), _rowvec(cls.reshape(-1), D, "cls"),
                                            D, L.ptr(out2), L.stream_ptr()), "vlm_image_rows_prep")
    return out2


def image_lead_rows(x, B, rows, lead):
    """x[b * rows] = lead for b < B (x: fp32 [B * rows, D] rows with unit column stride)."""
    L.require_cuda(x, lead)
    D = x.shape[1]
    if x.dtype != F32 or x.stride(1) != 1 or x.shape[0] != B * rows:
        raise L.VlmError("image_lead_rows: fp32 [B * rows, D]")
    L.check(L.get_lib().vlm_image_lead_rows(L.ptr(x), _ld(x), B, rows, D, _rowvec(lead, D, "lead"), L.stream_ptr()), "vlm_image_lead_rows")


def image_rows_bwd(g, B, rows, d_bias, d_type_row, d_cls):
    """g fp32 [B * rows, D] -> bf16 copy with the lead rows zeroed (returned); ADDS the column sums into d_bias (patch rows),
    d_type_row (all rows), d_cls (lead rows); None = not wanted."""
    L.require_cuda(g, d_bias, d_type_row, d_cls)
    n, D = g.shape
    if g.dtype != F32 or g.stride(1) != 1 or n != B * rows:
        raise L.VlmError("image_rows_bwd: fp32 [B * rows, D] gradient rows")
    g16 = torch.empty(n, D, device=g.device, dtype=BF16)
    L.check(L.get_lib().vlm_image_rows_bwd(L.ptr(g), _ld(g), B, rows, D, L.ptr(g16), _rowvec(d_bias, D, "d_bias"),
                                           _rowvec(d_type_row, D, "d_type_row"), _rowvec(d_cls.reshape(-1) if d_cls is not None else None, D, "d_cls"),
                                           L.ptr(_front_ws("image", D, g.device)), L.stream_ptr()), "vlm_image_rows_bwd")
    return g16


def tanh_fwd(x):
    """fp32 tanh of a bf16 [M, N] matrix with unit column stride."""
    L.require_cuda(x)
    M, N = x.shape
    if x.dtype != BF16 or x.stride(1) != 1:
        raise L.VlmError("tanh_fwd: bf16 [M, N] with unit column stride")
    y = torch.empty(M, N, device=x.device, dtype=F32)
    L.check(L.get_lib().vlm_tanh_fwd(L.ptr(x), _ld(x), M, N, L.ptr(y), L.stream_ptr()), "vlm_tanh_fwd")
    return y


ACT_BWD_GELU, ACT_BWD_TANH, ACT_BWD_NONE = 0, 1, 2


def act_bwd(g, saved, mode, Np=None):
    """bf16 [M, Np] (columns >= N zero) = g * act'(.): ACT_BWD_GELU from the saved bf16 pre-activation, ACT_BWD_TANH from the saved
    fp32 output, ACT_BWD_NONE: the cast / padding alone (saved = None)."""
    L.require_cuda(g, saved)
    M, N = g.shape
    Np = N if Np is None else Np
    if g.dtype not in (BF16, F32) or g.stride(1) != 1:
        raise L.VlmError("act_bwd: bf16 / fp32 gradient [M, N] with unit column stride")
    if mode != ACT_BWD_NONE:
        if saved.stride(1) != 1 or saved.shape[0] != M or saved.shape[1] < N:
            raise L.VlmError("act_bwd: saved [M, >= N] with unit column stride")
        if saved.dtype != (BF16 if mode == ACT_BWD_GELU else F32):
            raise L.VlmError("act_bwd: saved pre-activation is bf16 (GELU) / saved output is fp32 (tanh)")
    dy = torch.empty(M, Np, device=g.device, dtype=BF16)
    L.check(L.get_lib().vlm_act_bwd(L.ptr(g), int(g.dtype == F32), _ld(g), L.ptr(saved) if saved is not None else 0,
                                    _ld(saved) if saved is not None else 0, mode, M, N, Np, L.ptr(dy), L.stream_ptr()), "vlm_act_bwd")
    return dy


def colsum_small(a, out):
    """out[c] += sum_r a[r][c] for a bf16 matrix with at most 64 columns."""
    L.require_cuda(a, out)
    M, N = a.shape
    if a.dtype != BF16 or a.stride(1) != 1 or out.dtype != F32 or out.numel() < N or not out.is_contiguous():
        raise L.VlmError("colsum_small: bf16 [M, N <= 64], fp32 out")
    L.check(L.get_lib().vlm_colsum_small(L.ptr(a), _ld(a), M, N, L.ptr(out), L.stream_ptr()), "vlm_colsum_small")
    return out


def sample_negatives(sim_a, sim_b, B, u):
    """int64 [2, B]: row 0 drawn from softmax(sim_a[i, :]) without entry i, row 1 likewise from sim_b (fp32 [>= B, n], any strides);
    u: fp32 [2, B] uniforms."""
    L.require_cuda(sim_a, sim_b, u)
    n = sim_a.shape[1]
    for t in (sim_a, sim_b):
        if t.dtype != F32 or t.dim() != 2 or t.shape[0] < B or t.shape[1] != n:
            raise L.VlmError("sample_negatives: two fp32 similarity matrices [>= B, n]")
    if u.dtype != F32 or u.numel() != 2 * B or not u.is_contiguous():
        raise L.VlmError("sample_negatives: u is fp32 [2, B]")
    idx = torch.empty(2, B, device=sim_a.device, dtype=torch.int64)
    L.check(L.get_lib().vlm_sample_negatives(L.ptr(sim_a), sim_a.stride(0), sim_a.stride(1), L.ptr(sim_b), sim_b.stride(0), sim_b.stride(1), B, n,
                                             L.ptr(u), L.ptr(idx), L.stream_ptr()), "vlm_sample_negatives")
    return idx


def scatter_rows(R, D, dtype, sources, device):
    """[R, D] tensor of `dtype` (bf16 / fp32) = the sources placed at their rows, zero elsewhere.  sources: up to four
    (g [count, D] bf16 / fp32 with unit column stride, first_row, row_step)."""
    if dtype not in (BF16, F32) or len(sources) > 4:
        raise L.VlmError("scatter_rows: bf16 / fp32 output, at most four sources")
    dx = torch.empty(R, D, device=device, dtype=dtype)
    arr = (L.ScatterSrc * max(1, len(sources)))()
    for a, (g, first, step) in zip(arr, sources):
        L.require_cuda(g)
        if g.dim() != 2 or g.shape[1] != D or g.dtype not in (BF16, F32) or g.stride(1) != 1:
            raise L.VlmError("scatter_rows: sources are bf16 / fp32 [count, D] with unit column stride")
        if first + (g.shape[0] - 1) * step >= R and g.shape[0]:
            raise L.VlmError("scatter_rows: source rows %d + %d * j (j < %d) leave the %d output rows" % (first, step, g.shape[0], R))
        a.g, a.g_is_f32, a.ld, a.first_row, a.row_step, a.count = L.ptr(g), int(g.dtype == F32), _ld(g), first, step, g.shape[0]
    L.check(L.get_lib().vlm_scatter_rows(L.ptr(dx), int(dtype == F32), D, R, D, ctypes.cast(arr, ctypes.c_void_p), len(sources), L.stream_ptr()),
            "vlm_scatter_rows")
    return dx


def weighted_sum(terms, weights):
    """fp32 scalar = sum_k weights[k] * terms[k] (device scalars, host weights, at most 8)."""
    L.require_cuda(*terms)
    k = len(terms)
    if k == 0 or k > 8 or len(weights) != k or any(t.dtype != F32 or t.numel() != 1 for t in terms):
        raise L.VlmError("weighted_sum: 1..8 fp32 device scalars with one host weight each")
    out = torch.empty((), device=terms[0].device, dtype=F32)
    P = ctypes.c_void_p * k
    W = ctypes.c_float * k
    L.check(L.get_lib().vlm_weighted_sum(P(*[L.ptr(t) for t in terms]), W(*[float(w) for w in weights]), k, L.ptr(out), L.stream_ptr()),
            "vlm_weighted_sum")
    return out


def colsum(a, out):
    """out[n] += sum_m a[m,n]  (a bf16)."""
    L.require_cuda(a, out)
    M, N = a.shape
    L.check(L.get_lib().vlm_colsum_bf16(L.ptr(a), _ld(a), M, N, L.ptr(out), L.stream_ptr()), "vlm_colsum_bf16")
    return out


def adamw_step(p, g, m, v, p_bf16, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, zero_grad=True):
    """One fused AdamW update over flat fp32 buffers (HF-4.x semantics); `step` is the 1-based step count."""
    L.require_cuda(p, g, m, v, p_bf16)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr * (bc2 ** 0.5) / bc1
    rc = L.get_lib().vlm_adamw_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), L.ptr(p_bf16), p.numel(), lr, beta1, beta2,
                                    eps, weight_decay, step_size, grad_scale, int(zero_grad), L.stream_ptr())
    L.check(rc, "vlm_adamw_step")


def cast_bf16(src, dst):
    L.require_cuda(src, dst)
    L.check(L.get_lib().vlm_cast_f32_bf16(L.ptr(src), L.ptr(dst), src.numel(), L.stream_ptr()), "vlm_cast_f32_bf16")
    return dst


def transpose_table(pairs):
    """Device tile table for transpose_tiles: pairs = [(src bf16 [rows, cols] contiguous, dst bf16 [cols, rows]), ...]."""
    import numpy as np
    recs = []
    for src, dst in pairs:
        L.require_cuda(src, dst)
        if src.dtype != BF16 or dst.dtype != BF16 or not src.is_contiguous() or not dst.is_contiguous() \
                or src.dim() != 2 or tuple(dst.shape) != (src.shape[1], src.shape[0]):
            raise L.VlmError("transpose_table: contiguous bf16 [rows, cols] -> [cols, rows] pairs")
        rows, cols = src.shape
        for tr in range((rows + 63) // 64):
            for tc in range((cols + 63) // 64):
                recs.append((src.data_ptr(), dst.data_ptr(), rows | (cols << 32), tr | (tc << 32)))
    arr = np.array(recs, dtype=np.uint64).reshape(-1, 4)
    dev = pairs[0][0].device if pairs else "cuda"
    return torch.from_numpy(arr.view(np.int64)).to(dev), len(recs)


def transpose_tiles(table, n_tiles):
    """Refresh every transposed bf16 shadow listed in `table` (one launch), include/vlm_hip.h."""
    if n_tiles:
        L.check(L.get_lib().vlm_transpose_bf16_tiles(L.ptr(table), n_tiles, L.stream_ptr()), "vlm_transpose_bf16_tiles")


def droppath_sites(u0, u1, keeps, seq, out):
    """out[s][row] for every DropPath site s of a pass in one launch (include/vlm_hip.h vlm_droppath_sites)."""
    L.require_cuda(u0, u1, keeps, out)
    S = keeps.numel()
    if u0.dtype != F32 or keeps.dtype != F32 or out.dtype != F32 or tuple(u0.shape) != (S, seq.B) or not out.is_contiguous() \
            or out.shape[0] != S or (u1 is not None and tuple(u1.shape) != (S, seq.B)):
        raise L.VlmError("droppath_sites: u [S, B] f32, keeps [S] f32, out [S, rows] f32")
    L.check(L.get_lib().vlm_droppath_sites(L.ptr(u0), L.ptr(u1), L.ptr(keeps), S, seq.B, seq.n0, seq.n1, seq.base0, seq.base1,
                                           out.shape[1], L.ptr(out), L.stream_ptr()), "vlm_droppath_sites")
    return out


def droppath_rows(u, keep, seq, out):
    """out[row] = bernoulli(keep)/keep of the row's sample (u: one uniform draw per sample), include/vlm_hip.h."""
    L.require_cuda(u, out)
    if u.dtype != F32 or out.dtype != F32 or u.numel() < seq.B or not out.is_contiguous():
        raise L.VlmError("droppath_rows: u f32 [B], out contiguous f32 [rows]")
    L.check(L.get_lib().vlm_droppath_rows(L.ptr(u), float(keep), seq.B, seq.n0, seq.n1, seq.base0, seq.base1, L.ptr(out),
                                          L.stream_ptr()), "vlm_droppath_rows")
    return out


def gram_accumulate(x, gram64, tmp32=None):
    """gram64 (float64 [D,D]) += x^T x for bf16 or fp32 activations x [M,D] (the input of a hooked linear), in float64
    on the device (include/vlm_hip.h vlm_gram_f64; cache_gram_matrices.py:246-254)."""
    L.require_cuda(x, gram64)
    M, D = x.shape
    if gram64.dtype != torch.float64 or tuple(gram64.shape) != (D, D) or not gram64.is_contiguous():
        raise L.VlmError("gram accumulator must be a contiguous float64 [D,D] tensor")
    if x.dtype not in (BF16, F32):
        raise L.VlmError("gram_accumulate: bf16 or fp32 activations")
    L.check(L.get_lib().vlm_gram_f64(L.ptr(x), _ld(x), M, D, int(x.dtype == F32), L.ptr(gram64), L.stream_ptr()), "vlm_gram_f64")
    return gram64


F64 = torch.float64


def gemm_f64(a, b, c, ta=False, tb=False, alpha=1.0, beta=0.0):
    """c[M,N] = alpha * op(a) op(b) + beta * c in float64 on v_mfma_f64 (a: float64 or float32, b / c: float64)."""
    L.require_cuda(a, b, c)
    if b.dtype != F64 or c.dtype != F64 or a.dtype not in (F64, F32):
        raise L.VlmError("gemm_f64: a float64/float32, b and c float64")
    M, N = c.shape
    K = a.shape[0] if ta else a.shape[1]
    if (a.shape[1] if ta else a.shape[0]) != M or (b.shape[1] if tb else b.shape[0]) != K or (b.shape[0] if tb else b.shape[1]) != N:
        raise L.VlmError("gemm_f64: shapes %s %s -> %s" % (tuple(a.shape), tuple(b.shape), tuple(c.shape)))
    if K == 0:
        if beta == 0.0:
            c.zero_()
        return c
    L.check(L.get_lib().vlm_gemm_f64(int(ta), int(tb), M, N, K, float(alpha), L.ptr(a), _ld(a), int(a.dtype == F32), L.ptr(b),
                                     _ld(b), float(beta), L.ptr(c), _ld(c), L.stream_ptr()), "vlm_gemm_f64")
    return c


def scale_gram(src, dst, alpha, accumulate=False):
    """dst (+)= alpha * src + (1 - alpha) * diag(src), float64 [n,n] (vilt_module.py:388-392)."""
    L.require_cuda(src, dst)
    n = src.shape[0]
    if src.dtype != F64 or dst.dtype != F64 or tuple(src.shape) != (n, n) or tuple(dst.shape) != (n, n) \
            or not src.is_contiguous() or not dst.is_contiguous():
        raise L.VlmError("scale_gram: contiguous float64 [n,n] matrices")
    L.check(L.get_lib().vlm_scale_gram_f64(L.ptr(src), L.ptr(dst), n, float(alpha), int(accumulate), L.stream_ptr()),
            "vlm_scale_gram_f64")
    return dst


def cholesky_(s, status=None):
    """In-place lower Cholesky factor of the SPD float64 matrix s [n,n] (upper triangle left undefined): blocked
    right-looking factorisation, 64-wide block columns (potrf block, panel solve, MFMA-f64 trailing update), all block columns
    issued by ONE library call (vlm_cholesky_f64).  `status` (int32 [1] device tensor, zero on entry): when given, the
    positive-definiteness verdict is left there for the caller to read whenever it synchronises (several factorisations in
    flight on several streams); without it this call synchronises and raises on a non-positive pivot."""
    L.require_cuda(s, status)
    n = s.shape[0]
    if s.dtype != F64 or tuple(s.shape) != (n, n) or not s.is_contiguous():
        raise L.VlmError("cholesky_: contiguous float64 [n,n]")
    own = status is None
    if own:
        status = torch.zeros(1, device=s.device, dtype=torch.int32)
    L.check(L.get_lib().vlm_cholesky_f64(L.ptr(s), n, L.ptr(status), L.stream_ptr()), "vlm_cholesky_f64")
    if own:
        bad = int(status.item())
        if bad:
            raise L.VlmError("cholesky_: matrix is not positive definite (pivot %d)" % (bad - 1))
    return s


def solve_spd_right_(rhs, chol):
    """rhs [rows, n] <- rhs (L L^T)^-1 in place, `chol` from cholesky_: Y L^T = rhs forward over the block columns, then
    X L = Y backward (triangular block solves on the diagonal blocks, MFMA-f64 GEMMs for the off-diagonal updates), one
    library call (vlm_solve_spd_right_f64)."""
    L.require_cuda(rhs, chol)
    rows, n = rhs.shape
    if rhs.dtype != F64 or chol.dtype != F64 or tuple(chol.shape) != (n, n) or not chol.is_contiguous() or rhs.stride(1) != 1:
        raise L.VlmError("solve_spd_right_: float64 rhs [rows, n], contiguous float64 factor [n, n]")
    L.check(L.get_lib().vlm_solve_spd_right_f64(L.ptr(chol), n, L.ptr(rhs), _ld(rhs), rows, L.stream_ptr()),
            "vlm_solve_spd_right_f64")
    return rhs


F64_MAX_BATCH = 64


def _ptr_list(ts):
    arr = (ctypes.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = t.data_ptr()
    return arr


def gemm_f64_batched(a_list, b_list, c_list, alpha=1.0, beta=0.0):
    """c[i] = alpha * a[i] @ b[i] + beta * c[i] for products of ONE shape, one launch per 64 of them (vlm_gemm_f64_batched);
    a: contiguous float64 or float32 [M,K] (one dtype), b / c: contiguous float64."""
    if not (len(a_list) == len(b_list) == len(c_list)):
        raise L.VlmError("gemm_f64_batched: three lists of one length")
    if not a_list:
        return c_list
    L.require_cuda(*a_list, *b_list, *c_list)
    a0, b0, c0 = a_list[0], b_list[0], c_list[0]
    M, K = a0.shape
    N = b0.shape[1]
    for a, b, c in zip(a_list, b_list, c_list):
        if a.dtype != a0.dtype or a.dtype not in (F64, F32) or b.dtype != F64 or c.dtype != F64 or tuple(a.shape) != (M, K) \
                or tuple(b.shape) != (K, N) or tuple(c.shape) != (M, N) or not (a.is_contiguous() and b.is_contiguous() and c.is_contiguous()):
            raise L.VlmError("gemm_f64_batched: contiguous a [M,K] (one of float64/float32), float64 b [K,N], c [M,N] of one shape")
    if K == 0:
        if beta == 0.0:
            for c in c_list:
                c.zero_()
        return c_list
    for i in range(0, len(a_list), F64_MAX_BATCH):
        j = i + F64_MAX_BATCH
        L.check(L.get_lib().vlm_gemm_f64_batched(0, 0, M, N, K, float(alpha), _ptr_list(a_list[i:j]), K, int(a0.dtype == F32),
                                                 _ptr_list(b_list[i:j]), N, float(beta), _ptr_list(c_list[i:j]), N,
                                                 len(a_list[i:j]), L.stream_ptr()), "vlm_gemm_f64_batched")
    return c_list


def cholesky_batched_(mats, status):
    """cholesky_ of every matrix of `mats` (contiguous float64 [n, n], ONE n) in lock step: each block step is one launch over all
    of them (vlm_cholesky_f64_batched).  status: int32 device tensor [len(mats)], zero on entry."""
    L.require_cuda(status, *mats)
    n = mats[0].shape[0]
    for m in mats:
        if m.dtype != F64 or tuple(m.shape) != (n, n) or not m.is_contiguous():
            raise L.VlmError("cholesky_batched_: contiguous float64 [n,n] matrices of one size")
    for i in range(0, len(mats), F64_MAX_BATCH):
        chunk = mats[i:i + F64_MAX_BATCH]
        L.check(L.get_lib().vlm_cholesky_f64_batched(_ptr_list(chunk), len(chunk), n, L.ptr(status[i:i + len(chunk)]), L.stream_ptr()),
                "vlm_cholesky_f64_batched")
    return mats


def solve_spd_right_batched_(rhs, chols):
    """solve_spd_right_ for every (rhs[i], chols[i]) pair, all of ONE shape, in lock step (vlm_solve_spd_right_f64_batched)."""
    L.require_cuda(*rhs, *chols)
    rows, n = rhs[0].shape
    ld = _ld(rhs[0])
    for r, c in zip(rhs, chols):
        if r.dtype != F64 or c.dtype != F64 or tuple(r.shape) != (rows, n) or _ld(r) != ld or r.stride(1) != 1 \
                or tuple(c.shape) != (n, n) or not c.is_contiguous():
            raise L.VlmError("solve_spd_right_batched_: float64 rhs [rows, n] of one shape, contiguous float64 factors [n, n]")
    for i in range(0, len(rhs), F64_MAX_BATCH):
        rc, cc = rhs[i:i + F64_MAX_BATCH], chols[i:i + F64_MAX_BATCH]
        L.check(L.get_lib().vlm_solve_spd_right_f64_batched(_ptr_list(cc), n, _ptr_list(rc), ld, rows, len(rc), L.stream_ptr()),
                "vlm_solve_spd_right_f64_batched")
    return rhs


def patch_im2col(image, patches, patch, lead_rows):
    L.require_cuda(image, patches)
    B, C, H, W = image.shape
    if C != 3 or image.dtype != F32 or not image.is_contiguous():
        raise L.VlmError("patch_im2col expects a contiguous fp32 [B,3,H,W] image")
    rc = L.get_lib().vlm_patch_im2col(L.ptr(image), L.ptr(patches), B, H, W, patch, lead_rows, L.stream_ptr())
    L.check(rc, "vlm_patch_im2col")
    return patches


class Seq:
    """Segment-major token layout of one pass (include/vlm_hip.h): B samples, n0 text + n1 image tokens each."""

    __slots__ = ("B", "n0", "n1", "base0", "base1", "pos1")

    def __init__(self, B, n0, n1, base0=0, base1=None, pos1=None):
        self.B, self.n0, self.n1 = B, n0, n1
        self.base0 = base0
        self.base1 = base0 + B * n0 if base1 is None else base1
        self.pos1 = (n0 + 7) // 8 * 8 if pos1 is None else pos1  # image positions start at a multiple of 8

    @property
    def rows(self):
        return self.B * (self.n0 + self.n1)


def _attn_desc(qkv, seq, H, bias_t, head_row0, rel_index, rel_index_t, keep0, keep1, mode, scale, bias_dense=None):
    L.require_cuda(qkv, bias_t, rel_index, rel_index_t, keep0, keep1)
    if qkv.dtype != BF16:
        raise L.VlmError("attention expects bf16 qkv")
    d = L.AttnDesc()
    d.qkv = qkv.data_ptr()
    d.ld_qkv = _ld(qkv)
    d.H = H
    d.total_rows = qkv.shape[0]
    if bias_t is not None:
        if bias_t.dtype != F32 or not bias_t.is_contiguous():
            raise L.VlmError("attention bias_t must be contiguous f32")
        if rel_index_t is None or rel_index_t.dtype != torch.int16:
            raise L.VlmError("attention needs the transposed int16 relative index (rel_index_t)")
        d.R = bias_t.shape[1]
        d.bias_t = bias_t.data_ptr()
        d.rel_index_t = rel_index_t.data_ptr()
        d.ld_index_t = _ld(rel_index_t)
        d.index_t_rows = rel_index_t.shape[0]
        if rel_index is not None:
            if rel_index.dtype != torch.int16:
                raise L.VlmError("rel_index must be int16")
            d.rel_index = rel_index.data_ptr()
            d.ld_index = _ld(rel_index)
            d.index_rows = rel_index.shape[0]
    d.head_row0 = head_row0
    d.mode = mode
    for k in (keep0, keep1):
        if k is not None and (k.dtype != torch.uint8 or not k.is_contiguous()):
            raise L.VlmError("attention keep masks must be contiguous uint8")
    d.keep0 = keep0.data_ptr() if keep0 is not None else 0
    d.keep1 = keep1.data_ptr() if keep1 is not None else 0
    d.B, d.n0, d.n1, d.base0, d.base1, d.pos1 = seq.B, seq.n0, seq.n1, seq.base0, seq.base1, seq.pos1
    d.scale = scale
    if bias_t is not None:
        if bias_dense is None:
            if rel_index is None:
                raise L.VlmError("attention with a bias needs bias_dense (ops.bias_dense) or the int16 index")
            bias_dense = globals()["bias_dense"](bias_t, rel_index, seq, mode)  # one-off callers (tests)
        if not isinstance(bias_dense, DenseBias) or bias_dense.key != (seq.n0, seq.n1, seq.pos1, mode):
            raise L.VlmError("bias_dense must come from ops.bias_dense() for the same geometry and mode")
        d.bias_dense, d.bias_dense_t = bias_dense.q_major.data_ptr(), bias_dense.k_major.data_ptr()
        d.dense_tiles = bias_dense.tiles
        d._keep = bias_dense  # keep the tables alive until the launch is enqueued
    return d


class DenseBias:
    """fp16 relative-position bias of every (layer, head) in exponent units (bias * log2 e), tiled in MFMA operand order
    for ONE attention mode of one pass geometry, in both orientations (include/vlm_hip.h vlm_bias_dense; the
    reference's get_rel_pos_bias, vilt_module.py:1061-1064)."""

    __slots__ = ("q_major", "k_major", "tiles", "key")

    def __init__(self, q_major, k_major, tiles, key):
        self.q_major, self.k_major, self.tiles, self.key = q_major, k_major, tiles, key


def bias_dense(bias_t, index16, seq, mode):
    """Dense bias tables of a pass geometry `seq` (n0, n1, pos1) and attention mode, from the int16 index [NP, ld]."""
    L.require_cuda(bias_t, index16)
    if index16.dtype != torch.int16 or not index16.is_contiguous() or index16.shape[0] < seq.pos1 + seq.n1:
        raise L.VlmError("bias_dense: contiguous int16 index with pos1 + n1 rows")
    if bias_t.dtype != F32 or not bias_t.is_contiguous():
        raise L.VlmError("bias_dense: contiguous f32 bias_t [n_cols, R]")
    n_cols, R = bias_t.shape
    nbytes = L.get_lib().vlm_bias_dense_bytes(seq.n0, seq.n1, seq.pos1, mode)
    if nbytes == 0:
        raise L.VlmError("bias_dense: bad geometry (pos1 %% 8 == 0, pos1 >= n0)")
    outs = []
    for k_major in (0, 1):
        # 16 KiB of slack behind the last column: the attention streams request a block's operands up to a trip ahead, and the
        # last trip of the last (head, query block) asks for tiles behind the table (never used, but they must be mapped)
        flat = torch.empty(n_cols * (nbytes // 2) + 8192, device=bias_t.device, dtype=torch.float16)
        out = flat[: n_cols * (nbytes // 2)].view(n_cols, nbytes // 2)
        L.check(L.get_lib().vlm_bias_dense(L.ptr(bias_t), n_cols, R, L.ptr(index16), _ld(index16), seq.n0, seq.n1, seq.pos1,
                                           mode, k_major, L.ptr(out), L.stream_ptr()), "vlm_bias_dense")
        outs.append(out)
    return DenseBias(outs[0], outs[1], nbytes // 4096, (seq.n0, seq.n1, seq.pos1, mode))


def attention_fwd(qkv, out, lse, seq, H, *, bias_t=None, head_row0=0, rel_index=None, rel_index_t=None, keep0=None,
                  keep1=None, mode=L.ATTN_JOINT, scale=0.125, bias_dense=None):
    d = _attn_desc(qkv, seq, H, bias_t, head_row0, rel_index, rel_index_t, keep0, keep1, mode, scale, bias_dense)
    L.require_cuda(out, lse)
    rc = L.get_lib().vlm_attention_fwd(ctypes.byref(d), L.ptr(out), _ld(out), L.ptr(lse), L.stream_ptr())
    L.check(rc, "vlm_attention_fwd")
    return out


_ATTN_WS = {}


def attention_bwd(qkv, out, dout, lse, dqkv, seq, H, *, bias_t=None, head_row0=0, rel_index=None, rel_index_t=None,
                  keep0=None, keep1=None, mode=L.ATTN_JOINT, scale=0.125, dbias_t=None, delta_ws=None,
                  dq_colsum=None, dv_colsum=None, bias_dense=None):
    """dqkv <- d(loss)/d(qkv) (bf16, same layout as qkv); dbias_t += d(loss)/d(bias_t).  dq_colsum / dv_colsum:
    optional pairs (text-segment target, image-segment target) of f32 [H*64] vectors (None entries allowed) that
    receive += column sums of dQ / dV over that segment's rows (the q_bias / v_bias gradients)."""
    d = _attn_desc(qkv, seq, H, bias_t, head_row0, rel_index, rel_index_t, keep0, keep1, mode, scale, bias_dense)
    L.require_cuda(out, dout, lse, dqkv, dbias_t, delta_ws)
    cs = None
    if dq_colsum is not None or dv_colsum is not None:
        cs = L.AttnColsum()
        for name, pair in (("dq", dq_colsum), ("dv", dv_colsum)):
            for sgm, t in enumerate(pair or (None, None)):
                if t is not None:
                    L.require_cuda(t)
                    if t.dtype != F32 or t.numel() < H * 64 or not t.is_contiguous():
                        raise L.VlmError("attention_bwd: column-sum outputs must be contiguous f32 [H*64]")
                    getattr(cs, name)[sgm] = t.data_ptr()
    need = L.get_lib().vlm_attention_bwd_ws_floats(ctypes.byref(d), 0)  # (1: two-stage bias-gradient fold; measured no faster)
    if delta_ws is None or delta_ws.numel() < need:
        key = (qkv.device.index, torch.cuda.current_stream().cuda_stream)
        delta_ws = _ATTN_WS.get(key)
        if delta_ws is None or delta_ws.numel() < need:  # one scratch per stream, grown on demand
            delta_ws = _ATTN_WS[key] = torch.empty(need, device=qkv.device, dtype=F32)
    if bias_t is not None and rel_index is None:
        raise L.VlmError("attention_bwd needs both orientations of the int16 relative index")
    rc = L.get_lib().vlm_attention_bwd(ctypes.byref(d), L.ptr(out), _ld(out), L.ptr(dout), _ld(dout), L.ptr(lse),
                                       L.ptr(delta_ws), delta_ws.numel(), L.ptr(dqkv), _ld(dqkv), L.ptr(dbias_t),
                                       ctypes.byref(cs) if cs is not None else None, L.stream_ptr())
    L.check(rc, "vlm_attention_bwd")
    return dqkv


def debug_occupy(workgroups, threads, lds_bytes, microseconds):
    """Measurement tool: `workgroups` workgroups holding `lds_bytes` of LDS each spin for `microseconds` on the current stream
    (vlm_debug_occupy; ddp.FlatGradReducer(standin=...))."""
    L.check(L.get_lib().vlm_debug_occupy(int(workgroups), int(threads), int(lds_bytes), int(microseconds), L.stream_ptr()),
            "vlm_debug_occupy")


def _layerscale_jobs(jobs):
    """jobs: dicts with the fields of vlm_layerscale_job_t (tensors or None); yields ctypes arrays of <= 32 jobs."""
    for i in range(0, len(jobs), L.MAX_LAYERSCALE_JOBS):
        chunk = jobs[i:i + L.MAX_LAYERSCALE_JOBS]
        arr = (L.LayerScaleJob * len(chunk))()
        for a, j in zip(arr, chunk):
            w = j["weight"]
            L.require_cuda(w, *[v for v in j.values() if torch.is_tensor(v)])
            for f in ("weight", "gamma", "bias", "shadow", "bias_out", "raw_w", "raw_b", "dweight", "dbias", "dgamma"):
                t = j.get(f)
                setattr(a, f, t.data_ptr() if t is not None else None)
            a.N, a.K = w.shape[0], w.shape[1]
        yield arr, len(chunk)


def layerscale_fold(jobs):
    """W' = diag(gamma) W as the bf16 GEMM shadow, b' = gamma * b (vlm_layerscale_fold): dicts with weight, gamma, bias, shadow,
    bias_out."""
    for arr, n in _layerscale_jobs(jobs):
        L.check(L.get_lib().vlm_layerscale_fold(arr, n, L.stream_ptr()), "vlm_layerscale_fold")


def layerscale_finish(jobs):
    """Raw sums of the folded GEMMs -> gradients of W, b and gamma; the raw buffers are zeroed (vlm_layerscale_finish): dicts with
    weight, gamma, bias, raw_w, raw_b, dweight, dbias, dgamma."""
    for arr, n in _layerscale_jobs(jobs):
        L.check(L.get_lib().vlm_layerscale_finish(arr, n, L.stream_ptr()), "vlm_layerscale_finish")
