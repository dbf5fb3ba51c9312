"""Execution engine of the VLMo hot path on MI355X: flat parameter storage, the fused transformer-block
autograd function (hand-written forward AND backward over the HIP kernels), and small autograd wrappers for the
remaining GEMM / LayerNorm / patch-embed op sites.

Design (MI355X-first, not a port of the reference's module-by-module autograd):
  * every parameter is a view into ONE flat fp32 buffer; gradients, the bf16 GEMM shadows and the AdamW state
    live in parallel flat buffers with the same offsets -> zero_grad is one memset, the optimizer is one kernel
    per parameter group, a DDP bucket is a contiguous slice;
  * activations use the segment-major token layout of include/vlm_hip.h, so modality experts act on contiguous
    row ranges (no torch.cat / slicing copies per layer as in vision_transformer.py:554-603);
  * weight gradients are accumulated straight into the flat gradient buffer by the wgrad GEMM epilogue
    (weights are shared by up to six passes per training step).
"""
import itertools
import math
import os
import weakref
from typing import List, Optional

import torch

from . import _lib as L
from . import ops

BF16, F32 = torch.bfloat16, torch.float32
ALIGN = 64  # elements: keeps every parameter 256-B aligned in the fp32 and 128-B in the bf16 buffer


def _touch_hook(p):
    p._vlm_flat.touched.add(p._vlm_name)


def touch(*params):
    """Record that the running backward pass writes a gradient for these parameters (straight into the flat buffer)."""
    for p in params:
        if p is not None:
            f = getattr(p, "_vlm_flat", None)
            if f is not None:
                f.touched.add(p._vlm_name)


def _touch_expert(e):
    touch(e.n1w, e.n1b, e.qkvw, e.qb, e.vb, e.projw, e.projb, e.n2w, e.n2b, e.fc1w, e.fc1b, e.fc2w, e.fc2b)


_FLAT_SERIAL = itertools.count()


class FlatParams:
    """Flat fp32 master / grad / bf16-shadow storage for a module's parameters."""

    def __init__(self, module: torch.nn.Module, order_key=None):
        named = [(n, p) for n, p in module.named_parameters()]
        if order_key is not None:
            named.sort(key=lambda np_: order_key(np_[0]))
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.offsets = {}
        self.extent = {}  # padded footprint of a parameter in the flat buffers (what optimizer ranges / buckets cover)
        off = 0
        for i, (n, p) in enumerate(named):
            self.offsets[n] = (off, p.numel())
            ext = (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            # q_bias | zeros | v_bias laid out back to back: cat(q_bias, zeros_like(v_bias), v_bias)
            # (vision_transformer.py:335) becomes a VIEW of the flat buffer instead of two launches per block and pass.
            # The gap belongs to q_bias' footprint: zero parameters with zero gradients stay zero under AdamW.
            nxt = named[i + 1] if i + 1 < len(named) else None
            if (n.endswith(".q_bias") and nxt is not None and nxt[0] == n[:-len("q_bias")] + "v_bias"
                    and p.numel() % ALIGN == 0 and nxt[1].numel() == p.numel()):
                ext = 2 * p.numel()
                p._vlm_qkv_bias_span = 3 * p.numel()
            self.extent[n] = ext
            off += ext
        self.numel = off
        # names of the parameters a backward pass has written a gradient for so far: what torch calls `p.grad is not None`
        # (HF AdamW's `if p.grad is None: continue`, vilt_utils.py:314-317).  Kept structurally -- the engine's backward
        # functions call touch() on what they write, autograd-managed parameters report through a post-accumulate hook --
        # never by looking at gradient VALUES: a DropPath draw that drops a branch for the whole batch gives an exact-zero
        # gradient that the reference still treats as a gradient (weight decay applies)
        self.touched = set()
        self.serial = next(_FLAT_SERIAL)  # process-unique (id() of a freed FlatParams can come back)
        self.dense_cache = {}             # RelPos.dense_for: this model's dense bias tables, see make_relpos
        dev = self.params[0].device
        pad = 64 * 4096  # tail slack: K-strided GEMM operands may be addressed a few rows past a ragged weight
        self.flat_p = torch.zeros(off + pad, device=dev, dtype=F32)
        self.flat_g = torch.zeros(off + pad, device=dev, dtype=F32)
        self.flat_b = torch.zeros(off + pad, device=dev, dtype=BF16) if dev.type == "cuda" else None
        for n, p in named:
            o, k = self.offsets[n]
            self.flat_p[o:o + k].view_as(p).copy_(p.data)
            p.data = self.flat_p[o:o + k].view_as(p)
            p.grad = self.flat_g[o:o + k].view_as(p)
            p._vlm_name = n
            p._vlm_flat = self
            p.register_post_accumulate_grad_hook(_touch_hook)
            p._vlm_bf16 = self.flat_b[o:o + k].view_as(p) if self.flat_b is not None else None
            span = getattr(p, "_vlm_qkv_bias_span", 0)
            p._vlm_qkv_bias = self.flat_p[o:o + span] if span else None
        self.dirty = True
        # bumped whenever the fp32 masters change through this engine (optimizer step, shadow refresh after a load): what
        # derived tables (the dense relative-position bias) are cached on -- the HIP kernels write through raw pointers, so
        # torch's own tensor version counters do not see an optimizer step
        self.version = 0

    def refresh_shadow(self):
        """fp32 master -> bf16 GEMM operands (one cast kernel).  Called after load_state_dict / optimizer steps."""
        if self.flat_b is None:
            raise L.VlmError("the bf16 shadow lives on the GPU; move the model to cuda before flattening")
        ops.cast_bf16(self.flat_p[:self.numel], self.flat_b[:self.numel])
        self.refresh_transposed()
        self.dirty = False
        self.version += 1

    def enable_transposed(self, predicate):
        """Keep W^T (bf16, [in, out] row-major) next to the bf16 shadow of every 2-D parameter `predicate(name)` selects:
        the backward dX = dY . W then reads both GEMM operands K-contiguously (LDS-DMA) instead of staging the strided
        W through registers.  +2 B per selected weight; refreshed by ONE batched launch after each optimizer step."""
        if self.flat_b is None:
            return
        self.flat_bt = torch.zeros_like(self.flat_b)
        pairs = []
        for n, p in zip(self.names, self.params):
            if p.dim() == 2 and predicate(n):
                o, k = self.offsets[n]
                p._vlm_bf16_t = self.flat_bt[o:o + k].view(p.shape[1], p.shape[0])
                pairs.append((p._vlm_bf16, p._vlm_bf16_t))
        self._tr_table, self._tr_tiles = ops.transpose_table(pairs) if pairs else (None, 0)

    def refresh_transposed(self):
        self.refresh_folded()  # (the transposes are taken from the folded shadows)
        if getattr(self, "_tr_tiles", 0):
            ops.transpose_tiles(self._tr_table, self._tr_tiles)

    # ---- LayerScale folded into the branch's output projection (csrc/layerscale.hip) ----------------------------------------
    def enable_layerscale_fold(self, entries):
        """entries: (layer, weight, bias or None, gamma) per output projection whose branch is scaled by `gamma`
        (vision_transformer.py:489-491, :586, :603: attn.proj with gamma_1, mlp.fc2 with gamma_2, every expert).
        From now on the weight's bf16 shadows hold diag(gamma) W, `bias._vlm_folded` holds gamma * b, the wgrad GEMMs
        accumulate the RAW dL/dW' into `weight._vlm_raw` (and the row kernels colsum(g) into `bias._vlm_raw`), and
        finish_layerscale() turns the raw sums into the gradients of W, b and gamma."""
        if self.flat_b is None or not entries:
            return
        seen, total = set(), 0
        uniq = []
        for layer, w, b, g in entries:
            if id(w) in seen:
                continue
            seen.add(id(w))
            uniq.append((layer, w, b, g))
            total += (w.numel() + ALIGN - 1) // ALIGN * ALIGN + 2 * ((w.shape[0] + ALIGN - 1) // ALIGN * ALIGN)
        buf = torch.zeros(total, device=self.flat_p.device, dtype=F32)
        off = 0
        self._ls_jobs = {}

        def take(n, shape):
            nonlocal off
            t = buf[off:off + n].view(shape)
            off += (n + ALIGN - 1) // ALIGN * ALIGN
            return t

        for layer, w, b, g in uniq:
            w._vlm_raw = take(w.numel(), w.shape)
            raw_b = take(w.shape[0], (w.shape[0],))
            folded_b = take(w.shape[0], (w.shape[0],))
            if b is not None:
                b._vlm_raw, b._vlm_folded = raw_b, folded_b
            w._vlm_ls_gamma = g
            self._ls_jobs.setdefault(layer, []).append(dict(
                weight=w.data, gamma=g.data, bias=b.data if b is not None else None, shadow=w._vlm_bf16,
                bias_out=folded_b if b is not None else None, raw_w=w._vlm_raw, raw_b=raw_b if b is not None else None,
                dweight=w.grad, dbias=b.grad if b is not None else None, dgamma=g.grad))
        self._ls_buf = buf
        self.ls_pending = set()

    def shadow_reference(self):
        """What the bf16 shadow buffer must hold right now (tests): bf16(master) everywhere, bf16(diag(gamma) W) in the weights
        whose LayerScale is folded."""
        ref = self.flat_p[:self.numel].to(BF16)
        for jobs in getattr(self, "_ls_jobs", {}).values():
            for j in jobs:
                w = j["weight"]
                o = (w.data_ptr() - self.flat_p.data_ptr()) // 4
                ref[o:o + w.numel()] = (j["gamma"][:, None] * w).to(BF16).reshape(-1)
        return ref

    def refresh_folded(self):
        jobs = getattr(self, "_ls_jobs", None)
        if jobs:
            ops.layerscale_fold([j for layer in sorted(jobs) for j in jobs[layer]])

    def finish_layerscale(self, layer=None):
        """Raw sums -> gradients for `layer` (all pending layers when None), on the weight-gradient stream (behind the wgrad
        GEMMs that wrote the raw sums; the gradient reducer's buckets and the optimizer wait for that stream anyway)."""
        pend = getattr(self, "ls_pending", None)
        if not pend:
            return
        layers = sorted(pend) if layer is None else ([layer] if layer in pend else [])
        if not layers:
            return
        jobs = [j for i in layers for j in self._ls_jobs.get(i, ())]
        side = wgrad_stream()
        if side is None:
            ops.layerscale_finish(jobs)
        else:
            side.wait_stream(torch.cuda.current_stream())  # the row kernels' column sums (raw_b) come from the main stream
            with torch.cuda.stream(side):
                ops.layerscale_finish(jobs)
            if not _WGRAD["pending"]:  # outside a backward pass nobody is going to join the side stream for us
                torch.cuda.current_stream().wait_stream(side)
        for i in layers:
            pend.discard(i)

    def zero_grad(self):
        self.flat_g.zero_()
        # Folded LayerScale: in a normal step nothing is pending here (the end-of-backward callback, or the reducer right
        # before a bucket left, turned the raw sums into gradients and zeroed them).  A backward pass that RAISED (out of
        # memory inside a retry loop, a kernel error) leaves raw sums of the failed step behind and the callback latch set:
        # without this reset no later backward would arm its finish again and the stale sums would be added to the next step.
        if getattr(self, "ls_pending", None) or getattr(self, "_ls_armed", False):
            for jobs in getattr(self, "_ls_jobs", {}).values():
                for j in jobs:
                    j["raw_w"].zero_()
                    if j["raw_b"] is not None:
                        j["raw_b"].zero_()
            self.ls_pending.clear()
            self._ls_armed = False

    def slice_of(self, names):
        """(start, end) of the contiguous flat range covering `names` (a DDP bucket)."""
        lo = min(self.offsets[n][0] for n in names)
        hi = max(self.offsets[n][0] + self.extent[n] for n in names)
        return lo, hi


# ----------------------------------------------------------------------------------------------------------------
# weight-gradient side stream: wgrad GEMMs (a reduction over tokens into a small output: exactly one round of workgroups) are off
# the critical path of backward; issued on a second HIP stream the hardware co-schedules them with the dgrad / attention / row
# kernels of the main stream, whose last partial rounds leave CUs idle -- the more the smaller the pass: A/B on one box each,
# configs[4] (irtr, B = 20: M = 12 340 rows, 147 tiles of the N = 768 GEMMs on 256 CUs) 16.49 -> 15.58 ms per step (+5.8 %),
# configs[1] (ufo, B = 22) 71.59 -> 70.88 ms (+1.0 %).  The main stream re-joins at the end of backward (and the optimizer and
# every gradient bucket's collective wait for the side stream).  Default on since round 4 (the two-rank tests run with it);
# VLM_WGRAD_STREAM=0 keeps every wgrad on the main stream.
_WGRAD = {"stream": None, "enabled": os.environ.get("VLM_WGRAD_STREAM", "1") != "0", "pending": False}


def wgrad_stream():
    if not _WGRAD["enabled"]:
        return None
    if _WGRAD["stream"] is None:
        _WGRAD["stream"] = torch.cuda.Stream()
    return _WGRAD["stream"]


def sync_wgrad():
    """Make the current stream wait for every weight-gradient GEMM issued so far."""
    if _WGRAD["stream"] is not None:
        torch.cuda.current_stream().wait_stream(_WGRAD["stream"])
    _WGRAD["pending"] = False


def _arm_wgrad_join():
    if not _WGRAD["pending"]:
        _WGRAD["pending"] = True
        torch.autograd.Variable._execution_engine.queue_callback(sync_wgrad)


class _Side:
    """with _Side(tensors...): run the enclosed launches on the wgrad stream after everything issued so far."""

    def __init__(self, *tensors):
        self.s = wgrad_stream()
        self.tensors = tensors
        self.ctx = None

    def __enter__(self):
        if self.s is None:
            return self
        self.s.wait_stream(torch.cuda.current_stream())
        for t in self.tensors:
            t.record_stream(self.s)
        _arm_wgrad_join()
        self.ctx = torch.cuda.stream(self.s)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.ctx is not None:
            self.ctx.__exit__(*a)
        return False


# ----------------------------------------------------------------------------------------------------------------
# modality experts on two streams: in an all_moe block the text rows (B*40) and the image rows (B*577) go through DIFFERENT
# weights, so each expert's chain (LayerNorm -> GEMM ...) is an independent sequence of launches.  The text GEMMs are small
# (M = 3 520: 168-504 tiles of 128x128 on 512 slots) and ran at a third of the image GEMMs' rate, 9.6 % of the all_moe step for
# 6.5 % of its rows; issued on a second HIP stream they fill the CUs the image expert's last partial round of tiles leaves
# idle.  Fork / join per phase (VLM_EXPERT_STREAMS=0 switches it off).
_EXPERTS = {"stream": None, "enabled": os.environ.get("VLM_EXPERT_STREAMS", "1") != "0"}


# Grouped GEMM (default): the experts' forward and dgrad GEMMs of a block are ONE launch each over both row
# ranges (ops.gemm_grouped): the text expert's 14 row tiles ride in the image expert's rounds of 256x256 tiles instead of
# under-filling launches of their own on the 128x128 kernel (9.6 % of the all_moe step for 6.5 % of its rows).  The two-stream
# schedule above is then off (every GEMM is a join).  VLM_GROUPED_GEMM=0: back to one launch chain per expert.
_GROUPED = os.environ.get("VLM_GROUPED_GEMM", "1") != "0"


def _use_grouped(ranges):
    return _GROUPED and len(ranges) > 1 and all(wT16(getattr(e, n)) is not None for _, _, e in ranges
                                                for n in ("qkvw", "projw", "fc1w", "fc2w"))


class _ExpertStreams:
    def __init__(self, ranges):
        self.side = None
        self.small = -1
        if _EXPERTS["enabled"] and len(ranges) > 1 and not _use_grouped(ranges):
            if _EXPERTS["stream"] is None:
                _EXPERTS["stream"] = torch.cuda.Stream()
            self.side = _EXPERTS["stream"]
            sizes = [r1 - r0 for r0, r1, _ in ranges]
            self.small = sizes.index(min(sizes))  # the smallest row range (text) goes to the side stream

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(torch.cuda.current_stream())
        return self

    def on(self, idx):
        if self.side is not None and idx == self.small:
            return torch.cuda.stream(self.side)
        return _NULL_CTX

    def __exit__(self, *a):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        return False


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()


def wT16(p):
    """Transposed bf16 shadow [in, out] of a weight, or None when the model keeps none for it."""
    return getattr(p, "_vlm_bf16_t", None)


def _dgrad(dy, w, dx, **epi):
    """dx = dy . W (+ epilogue): through W^T as a K-contiguous GEMM when the transposed shadow exists."""
    wt = wT16(w)
    if wt is not None:
        return ops.gemm(dy, wt, dx, **epi)
    return ops.gemm(dy, w16(w), dx, tb=True, **epi)


def w16(p):
    """bf16 shadow of a parameter (fails loudly if the model was not flattened onto the GPU)."""
    s = getattr(p, "_vlm_bf16", None)
    if s is None:
        raise L.VlmError("parameter has no bf16 shadow: call model.setup_engine() on a CUDA model first")
    return s


# ----------------------------------------------------------------------------------------------------------------
# relative-position bias handle
class RelPos:
    """What the attention kernel needs instead of the reference's dense [H*L, N, N] bias (vilt_module.py:1061):
    the transposed table (autograd-connected), the int16 index in index coordinates and its transpose."""

    def __init__(self, bias_t, index16, index16_t, holder, cache_tag=None, cache=None):
        self.bias_t = bias_t          # [H*L, R] fp32, requires grad in training
        self.index = index16          # int16 [NP, ld]
        self.index_t = index16_t
        self.holder = holder          # _TableT ctx holder: accumulates d(bias_t)
        self._dense = {}              # (n0, n1, pos1, mode) -> ops.DenseBias, built on first use inside the pass
        self.cache_tag = cache_tag    # identifies (table contents, index) across passes, or None: no cross-pass cache
        self.cache = cache            # the owning FlatParams' dict (dies with the model), or None

    def dense_for(self, seq, mode):
        """Tiled fp16 bias of all layers and heads for this pass geometry and attention mode (2 launches of vlm_bias_dense):
        the attention kernels add their (layer, head) slice to the score accumulators on the matrix pipe.  Built once per
        pass geometry and TABLE VERSION: a training step changes the table, so it rebuilds once per step and pass (the
        minimum); a no_grad sweep over many batches (compute_irtr_recall: ~250 passes over one table) builds it once."""
        key = (seq.n0, seq.n1, seq.pos1, mode)
        d = self._dense.get(key)
        if d is None:
            ck = self.cache_tag + key if self.cache_tag is not None and self.cache is not None else None
            cache = self.cache
            d = cache.get(ck) if ck is not None else None
            if d is None:
                with torch.no_grad():
                    d = ops.bias_dense(self.bias_t.detach(), self.index, seq, mode)
                _DENSE_STATS["built"] += 1
                if ck is not None:
                    for k in [k for k in cache if k[:2] != ck[:2]]:  # older contents of this model's table
                        del cache[k]
                    while len(cache) >= 8:
                        del cache[next(iter(cache))]
                    cache[ck] = d
            else:
                _DENSE_STATS["hits"] += 1
            self._dense[key] = d
        return d

    @property
    def dbias_t(self):
        return self.holder["dbias_t"]


_ZERO = {}


def _zero_scalar(dev):
    """A constant fp32 zero on `dev` (never written): the placeholder gradient autograd needs to reach _TableT.backward."""
    z = _ZERO.get(str(dev))
    if z is None:
        z = _ZERO[str(dev)] = torch.zeros((), device=dev, dtype=F32)
    return z


class _TableT(torch.autograd.Function):
    """bias_t = table^T (contiguous).  Backward adds the gradient the attention kernels accumulated straight into the table's
    slice of the flat gradient buffer.  The transpose is taken once per (optimizer step, in-place edit) and shared by the passes
    of a step (round 6: one copy instead of one per pass)."""

    @staticmethod
    def forward(ctx, table, holder):
        ctx.holder, ctx.table = holder, table
        flat = getattr(table, "_vlm_flat", None)
        if flat is None or flat.dirty:
            return table.detach().t().contiguous()
        key = (flat.version, table._version, table.data_ptr())
        ent = getattr(flat, "table_t", None)
        if ent is None or ent[0] != key:
            ent = flat.table_t = (key, table.detach().t().contiguous())
        return ent[1].view_as(ent[1])  # a fresh tensor object per pass over the shared storage (nobody writes bias_t)

    @staticmethod
    def backward(ctx, g):
        acc = ctx.holder.pop("dbias_t", None)
        ctx.holder.pop("routed", None)
        table = ctx.table
        placeholder = g.stride() == (0,) * g.dim()  # _BlockFn's expanded zero: carries nothing
        if acc is None:
            return (None if placeholder else g.t()), None
        if getattr(table, "_vlm_flat", None) is not None and table.requires_grad and table.grad is not None:
            touch(table)
            table.grad.add_(acc.t() if placeholder else (acc + g).t())
            return None, None
        return (acc if placeholder else acc + g).t(), None


# The cross-pass cache of dense bias tables lives ON the model's FlatParams (`flat.dense_cache`: it dies with the model and a
# rebuilt model starts empty -- a module-global keyed by id() / data_ptr() could hit on a recycled id after a model was freed),
# keyed by (flat.version, table._version, index tag, n0, n1, pos1, mode) -> ops.DenseBias.
_DENSE_STATS = {"built": 0, "hits": 0}


def make_relpos(table, index16, index16_t, index_tag=None):
    """`index_tag` names the index buffer for the cross-pass cache (the model's own buffers: their _idx_cache key); None -- a
    temporary index whose storage the allocator may hand to a different index next -- switches the cache off."""
    holder = {}
    bias_t = _TableT.apply(table, holder)
    if torch.is_grad_enabled() and table.requires_grad:
        holder["dbias_t"] = torch.zeros_like(bias_t)
    flat = getattr(table, "_vlm_flat", None)
    # flat.version follows optimizer steps / reloads (raw-pointer writes), table._version follows in-place torch edits;
    # a model whose masters are marked dirty has pending edits: no cache for that pass
    ok = flat is not None and not flat.dirty and index_tag is not None
    tag = (flat.version, table._version, index_tag) if ok else None
    return RelPos(bias_t, index16, index16_t, holder, tag, flat.dense_cache if ok else None)


# ----------------------------------------------------------------------------------------------------------------
_KEEPS_CACHE = {}


class PassCtx:
    """Static description of one pass (one `infer*` call) shared by its 12(+2) block evaluations."""

    def __init__(self, seq: ops.Seq, num_heads: int, relpos: Optional[RelPos], keep0=None, keep1=None):
        self.seq = seq
        self.H = num_heads
        self.relpos = relpos
        self.keep0 = keep0
        self.keep1 = keep1
        self.gram = None  # GramCapture when the Gram cache is being recorded
        self._row2sample = None
        self._dp_pool, self._dp_next = None, 0
        self._dp_sites, self._dp_all, self._dp_call = None, None, 0
        self.independent_segments = False  # True: the two segments are separate passes of the reference (own DropPath draws)
        self.uniform_source = None  # callable(pass_ctx, n_sites, n_streams) -> uniforms; None = torch.rand

    def row2sample(self, device):
        if self._row2sample is None:
            s = self.seq
            a = torch.arange(s.B, device=device).repeat_interleave(s.n0)
            b = torch.arange(s.B, device=device).repeat_interleave(s.n1)
            self._row2sample = torch.cat([a, b])
        return self._row2sample

    def plan_drop_path(self, probs):
        """Tell the pass the DropPath probabilities of its sites in call order (two per block evaluation): all row
        scales are then produced by ONE launch (two for a unimodal pair) at the first site instead of one per site."""
        self._dp_sites = [float(p) for p in probs]

    def drop_path_rows(self, prob, training, device):
        """Per-row scale of timm's DropPath (per-sample bernoulli(keep)/keep), or None when inactive."""
        if not training or prob <= 0.0:
            self._dp_call += 1
            return None
        s = self.seq
        rows = max(s.base0 + s.B * s.n0, s.base1 + s.B * s.n1)
        i = self._dp_call
        self._dp_call += 1
        if self._dp_sites is not None and i < len(self._dp_sites) and abs(self._dp_sites[i] - prob) < 1e-12:
            if self._dp_all is None:
                S = len(self._dp_sites)
                key = (tuple(self._dp_sites), str(device))
                keeps = _KEEPS_CACHE.get(key)
                if keeps is None:  # uploaded once per schedule: a pageable H2D copy inside a step would stall the queue
                    keeps = _KEEPS_CACHE[key] = torch.tensor([1.0 - p if p > 0.0 else 1.0 for p in self._dp_sites],
                                                             dtype=F32).to(device)
                streams = 2 if (self.independent_segments and s.n0 and s.n1) else 1
                if self.uniform_source is not None:
                    # injected draws (parity tests): fp32 [streams, S, B] in [0, 1); a sample's branch is kept where
                    # u < keep.  Stream 0 drives segment 0 (text rows), stream 1 segment 1 (image rows).
                    u = self.uniform_source(self, S, streams).to(device=device, dtype=F32).contiguous()
                    assert u.shape == (streams, S, s.B), (u.shape, streams, S, s.B)
                else:
                    u = torch.rand(streams, S, s.B, device=device, dtype=F32)
                self._dp_all = ops.droppath_sites(u[0], u[1] if u.shape[0] == 2 else None, keeps, s,
                                                  torch.empty(S, rows, device=device, dtype=F32))
            return self._dp_all[i]
        # unplanned site: one launch (two for independent segments) from a pooled uniform draw
        if self._dp_pool is None or self._dp_next + 2 > self._dp_pool.shape[0]:
            self._dp_pool = torch.rand(64, s.B, device=device, dtype=F32)
            self._dp_next = 0
        u = self._dp_pool[self._dp_next]
        self._dp_next += 1
        out = torch.empty(rows, device=device, dtype=F32)
        if self.independent_segments and s.n0 and s.n1:
            u1 = self._dp_pool[self._dp_next]
            self._dp_next += 1
            ops.droppath_rows(u, 1.0 - prob, ops.Seq(s.B, s.n0, 0, base0=s.base0, base1=s.base1), out)
            return ops.droppath_rows(u1, 1.0 - prob, ops.Seq(s.B, 0, s.n1, base0=s.base0, base1=s.base1), out)
        return ops.droppath_rows(u, 1.0 - prob, s, out)


class ExpertWeights:
    """Pointers one modality expert contributes to a block evaluation (all tensors are parameters)."""

    __slots__ = ("n1w", "n1b", "qkvw", "qb", "vb", "projw", "projb", "n2w", "n2b", "fc1w", "fc1b", "fc2w", "fc2b",
                 "gram_names")


class GramCapture:
    """Device-resident Gram accumulators of the inputs of the hooked linears (cache_gram_matrices.py:246-281):
    name -> float64 [D,D].  `names(expert)` maps an expert to its four module names (qkv's key is the Attention
    module, the others are the Linear modules), or fewer when a module is not hooked in that architecture."""

    def __init__(self):
        self.grams = {}

    def add(self, name, x):
        """G[name] += x^T x in float64 on the device (v_mfma_f64 SYRK, ops.gram_accumulate); x: bf16 or fp32 [rows, D]."""
        if name is None or x.shape[0] == 0:
            return
        D = x.shape[1]
        g = self.grams.get(name)
        if g is None:
            g = self.grams[name] = torch.zeros(D, D, device=x.device, dtype=torch.float64)
        ops.gram_accumulate(x if x.stride(1) == 1 else x.contiguous(), g)

    def all_reduce(self, group=None):
        """Sum every accumulator over the ranks of a data-parallel evaluation (the reference's hooks run
        under the Trainer on every rank and only rank-local sums reach its torch.save; here the float64 matrices are
        all-reduced on the device -- one collective per matrix, same key set on every rank)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return self
        # a rank that was dealt no batch (more ranks than batches) has hooked nothing: agree on the key set first
        mine = {k: int(v.shape[0]) for k, v in self.grams.items()}
        every = [None] * dist.get_world_size(group)
        dist.all_gather_object(every, mine, group=group)
        dev = next(iter(self.grams.values())).device if self.grams else torch.device("cuda", torch.cuda.current_device())
        for d in every:
            for k, D in d.items():
                if k not in self.grams:
                    self.grams[k] = torch.zeros(D, D, device=dev, dtype=torch.float64)
        for k in sorted(self.grams):
            dist.all_reduce(self.grams[k], group=group)
        return self

    def state_dict(self):
        """What the reference torch.save()s: name -> float64 CPU tensor (cache_gram_matrices.py:349)."""
        return {k: v.cpu() for k, v in self.grams.items()}


# A/B switch for measurements: 0 = separate colsum launches, 1/2 = q/v bias inside the attention backward and fc1 bias
# through the GELU-backward GEMM epilogue + fold workspace
_DEFER_FOLD = os.environ.get("VLM_DEFER_FOLD", "1") != "0"
_DENSE_BIAS = True  # the attention kernels always read the dense table (round 2: bias enters through the matrix pipe)
_SAVE_DERIV = os.environ.get("VLM_GELU_SAVE_DERIV", "1") != "0"  # fc1 saves gelu'(h) instead of h for the backward pass
_FUSE_MODE = int(os.environ.get("VLM_FUSE_BIAS_GRADS", "2"))
_FUSE_BIAS_GRADS = _FUSE_MODE != 0
_FUSE_FC1_BIAS = _FUSE_MODE in (1, 2)  # through the fold workspace (mode 0: separate colsum launch)


def _segment_bias_grads(ranges, seq):
    """Map the two token segments of a pass to the q_bias / v_bias gradient vectors of the expert that owns them."""
    qb, vb = [None, None], [None, None]
    for sgm, (base, n) in enumerate(((seq.base0, seq.n0), (seq.base1, seq.n1))):
        if n == 0:
            continue
        lo, hi = base, base + seq.B * n
        owner = [e for r0, r1, e in ranges if r0 <= lo and hi <= r1]
        if len(owner) != 1:
            return None, None, False
        if owner[0].qb is not None:
            qb[sgm], vb[sgm] = owner[0].qb.grad, owner[0].vb.grad
    return qb, vb, True


class BlockPlan:
    """Routing of one block evaluation: row ranges -> experts, attention mode.  Built by Block.plan()."""

    def __init__(self, ranges, mode, gamma1, gamma2, layer, drop_prob, eps):
        self.ranges = ranges      # list of (r0, r1, ExpertWeights)
        self.mode = mode
        self.gamma1, self.gamma2 = gamma1, gamma2
        self.layer = layer
        self.drop_prob = drop_prob
        self.eps = eps


def _qkv_bias(e):
    # vision_transformer.py:335: cat(q_bias, zeros_like(v_bias), v_bias)
    if e.qb is None:
        return None
    view = getattr(e.qb, "_vlm_qkv_bias", None)
    if view is not None:
        return view  # [q_bias | 0 | v_bias] as laid out by FlatParams
    return torch.cat((e.qb.detach(), torch.zeros_like(e.vb), e.vb.detach()))


# norm2's backward and the attention branch's LayerScale backward as ONE pass over the rows (K3-style fusion of two row kernels:
# the second one's 4-B-per-element read of the residual-stream gradient and its launch go; bit-identical).  VLM_FUSE_LN_SCALE=0:
# two launches.
_FUSE_LN_SCALE = os.environ.get("VLM_FUSE_LN_SCALE", "1") != "0"


# LayerScale folded into attn.proj / mlp.fc2 (csrc/layerscale.hip, FlatParams.enable_layerscale_fold): the residual epilogues of
# the two forward GEMMs lose their column scale and their bf16 copy of the branch output, the LayerScale backward shrinks to a
# cast (+ column sums), and dgamma comes from the weight gradient.  VLM_FOLD_LAYERSCALE=0: the round-1..4 form (A/B).
_FOLD_LS = os.environ.get("VLM_FOLD_LAYERSCALE", "1") != "0"


def _folded(ranges):
    return _FOLD_LS and all(getattr(e.projw, "_vlm_raw", None) is not None and getattr(e.fc2w, "_vlm_raw", None) is not None
                            for _, _, e in ranges)


def _fb(b):
    return b._vlm_folded if b is not None else None


def _rb(b):
    return b._vlm_raw if b is not None else None


def _arm_layerscale_finish(flat):
    """End of the running backward pass: raw sums -> gradients for every block the pass touched (a block whose bucket the
    gradient reducer has sent already was finished there)."""
    if getattr(flat, "_ls_armed", False):
        return
    flat._ls_armed = True

    def done():
        flat._ls_armed = False
        flat.finish_layerscale()
        sync_wgrad()  # the gradients are written on the side stream: whoever reads .grad next on this stream sees them

    torch.autograd.Variable._execution_engine.queue_callback(done)


def _ln2_bwd_scale1(dln, x1, st2, e, dx1, dx2, y1, g1, rs1, dy1, fold):
    if y1 is None:  # folded LayerScale: dy1 = bf16(rs1 * dx1) and its column sums (the raw proj-bias gradient)
        if _FUSE_LN_SCALE and x1.shape[1] <= 768:
            ops.layernorm_bwd_scale(dln, x1, st2, e.n2w, dx1, dx2, e.n2w.grad, e.n2b.grad, y=None, sgamma=None, row_scale=rs1,
                                    sdy=dy1, dsgamma=None, dsbias=_rb(e.projb), fold=fold)
        else:
            ops.layernorm_bwd(dln, x1, st2, e.n2w, dx1, dres=dx2, dgamma=e.n2w.grad, dbeta=e.n2b.grad, fold=fold)
            ops.layerscale_bwd(dx1, None, None, rs1, dy1, None, _rb(e.projb), fold=fold)
        return
    if _FUSE_LN_SCALE and x1.shape[1] <= 768:  # (wider rows: the fused kernel's four accumulator sets spill)
        ops.layernorm_bwd_scale(dln, x1, st2, e.n2w, dx1, dx2, e.n2w.grad, e.n2b.grad, y=y1, sgamma=g1, row_scale=rs1, sdy=dy1,
                                dsgamma=g1.grad, dsbias=e.projb.grad, fold=fold)
    else:
        ops.layernorm_bwd(dln, x1, st2, e.n2w, dx1, dres=dx2, dgamma=e.n2w.grad, dbeta=e.n2b.grad, fold=fold)
        ops.layerscale_bwd(dx1, y1, g1, rs1, dy1, g1.grad, e.projb.grad, fold=fold)


class _BlockFn(torch.autograd.Function):
    """One transformer block evaluation (LayerNorm -> QKV -> attention -> proj+LayerScale+residual -> LayerNorm ->
    fc1+GELU -> fc2+LayerScale+residual), forward and hand-written backward over the HIP kernels.
    Reference: Block.plain_forward / separate_plain_forward / moe_forward, vision_transformer.py:525-681."""

    @staticmethod
    def forward(ctx, x, bias_t, plan: BlockPlan, pc: PassCtx, training: bool, hook):
        M, D = x.shape
        dev = x.device
        H = pc.H
        Fdim = plan.ranges[0][2].fc1w.shape[0]
        x = x.contiguous()
        ln1 = torch.empty(M, D, device=dev, dtype=BF16)
        st1 = torch.empty(M, 2, device=dev, dtype=F32)
        qkv = torch.empty(M, 3 * D, device=dev, dtype=BF16)
        rs1 = pc.drop_path_rows(plan.drop_prob, training, dev)
        rs2 = pc.drop_path_rows(plan.drop_prob, training, dev)
        grouped = _use_grouped(plan.ranges)
        if grouped:
            for r0, r1, e in plan.ranges:
                ops.layernorm_fwd(x[r0:r1], e.n1w, e.n1b, plan.eps, ln1[r0:r1], st1[r0:r1])
            ops.gemm_grouped(ln1, [(r0, r1, w16(e.qkvw), _qkv_bias(e), None) for r0, r1, e in plan.ranges], qkv)
        else:
            with _ExpertStreams(plan.ranges) as es:
                for idx, (r0, r1, e) in enumerate(plan.ranges):
                    with es.on(idx):
                        ops.layernorm_fwd(x[r0:r1], e.n1w, e.n1b, plan.eps, ln1[r0:r1], st1[r0:r1])
                        ops.gemm(ln1[r0:r1], w16(e.qkvw), qkv[r0:r1], bias=_qkv_bias(e))
        o = torch.empty(M, D, device=dev, dtype=BF16)
        lse = torch.empty(H, M, device=dev, dtype=F32)
        rp = pc.relpos
        ops.attention_fwd(qkv, o, lse, pc.seq, H, bias_t=bias_t, head_row0=plan.layer * H,
                          rel_index=rp.index if rp is not None else None,
                          rel_index_t=rp.index_t if rp is not None else None, keep0=pc.keep0, keep1=pc.keep1,
                          mode=plan.mode, bias_dense=rp.dense_for(pc.seq, plan.mode) if rp is not None else None)
        x1 = torch.empty(M, D, device=dev, dtype=F32)
        folded = _folded(plan.ranges)
        # folded: W' = diag(gamma) W and b' = gamma * b are the GEMM operands -- no column scale, no saved branch output
        y1 = torch.empty(M, D, device=dev, dtype=BF16) if not folded else None
        ln2 = torch.empty(M, D, device=dev, dtype=BF16)
        st2 = torch.empty(M, 2, device=dev, dtype=F32)
        h = torch.empty(M, Fdim, device=dev, dtype=BF16)
        a = torch.empty(M, Fdim, device=dev, dtype=BF16)
        x2 = torch.empty(M, D, device=dev, dtype=F32)
        y2 = torch.empty(M, D, device=dev, dtype=BF16) if not folded else None
        cs1, cs2 = (None, None) if folded else (plan.gamma1, plan.gamma2)
        pb = (lambda e: _fb(e.projb)) if folded else (lambda e: e.projb)
        fb2 = (lambda e: _fb(e.fc2b)) if folded else (lambda e: e.fc2b)
        if grouped:
            rg = plan.ranges
            ops.gemm_grouped(o, [(r0, r1, w16(e.projw), pb(e), None) for r0, r1, e in rg], x1, col_scale=cs1,
                             row_scale=rs1, residual=x, aux=y1)
            for r0, r1, e in rg:
                ops.layernorm_fwd(x1[r0:r1], e.n2w, e.n2b, plan.eps, ln2[r0:r1], st2[r0:r1])
            ops.gemm_grouped(ln2, [(r0, r1, w16(e.fc1w), e.fc1b, None) for r0, r1, e in rg], a,
                             act=L.ACT_GELU_DERIV if _SAVE_DERIV else L.ACT_GELU, aux=h)
            ops.gemm_grouped(a, [(r0, r1, w16(e.fc2w), fb2(e), None) for r0, r1, e in rg], x2, col_scale=cs2,
                             row_scale=rs2, residual=x1, aux=y2)
        else:
            with _ExpertStreams(plan.ranges) as es:  # each expert's proj -> LayerNorm -> fc1 -> fc2 chain is independent
                for idx, (r0, r1, e) in enumerate(plan.ranges):
                    with es.on(idx):
                        ops.gemm(o[r0:r1], w16(e.projw), x1[r0:r1], bias=pb(e), col_scale=cs1,
                                 row_scale=rs1[r0:r1] if rs1 is not None else None, residual=x[r0:r1],
                                 aux=y1[r0:r1] if y1 is not None else None)
                        ops.layernorm_fwd(x1[r0:r1], e.n2w, e.n2b, plan.eps, ln2[r0:r1], st2[r0:r1])
                        # h = gelu'(pre-activation) (VLM_GELU_SAVE_DERIV, default): the forward epilogue has erf and exp in
                        # hand anyway, and the fc2-dgrad epilogue of the backward pass becomes a multiplication
                        ops.gemm(ln2[r0:r1], w16(e.fc1w), a[r0:r1], bias=e.fc1b,
                                 act=L.ACT_GELU_DERIV if _SAVE_DERIV else L.ACT_GELU, aux=h[r0:r1])
                        ops.gemm(a[r0:r1], w16(e.fc2w), x2[r0:r1], bias=fb2(e), col_scale=cs2,
                                 row_scale=rs2[r0:r1] if rs2 is not None else None, residual=x1[r0:r1],
                                 aux=y2[r0:r1] if y2 is not None else None)
        if pc.gram is not None:
            # Gram cache: the LayerNorm outputs enter in fp32 (re-computed here, capture runs are not timed), like the
            # fp32 activations the reference's hooks see; the attention output and the GELU output exist only as the bf16
            # tensors the next GEMM consumes
            for r0, r1, e in plan.ranges:
                gn = e.gram_names
                if gn.get("qkv") or gn.get("fc1"):
                    t32 = torch.empty(r1 - r0, D, device=dev, dtype=F32)
                    st = torch.empty(r1 - r0, 2, device=dev, dtype=F32)
                    if gn.get("qkv"):
                        ops.layernorm_fwd(x[r0:r1], e.n1w, e.n1b, plan.eps, t32, st)
                        pc.gram.add(gn.get("qkv"), t32)
                    if gn.get("fc1"):
                        ops.layernorm_fwd(x1[r0:r1], e.n2w, e.n2b, plan.eps, t32, st)
                        pc.gram.add(gn.get("fc1"), t32)
                pc.gram.add(gn.get("proj"), o[r0:r1])
                pc.gram.add(gn.get("fc2"), a[r0:r1])
        ctx.plan, ctx.pc, ctx.hook = plan, pc, hook
        ctx.has_bias = bias_t is not None
        ctx.folded = folded
        ctx.save_for_backward(x, st1, ln1, qkv, o, lse, y1 if y1 is not None else x.new_empty(0), x1, st2, ln2, h, a,
                              y2 if y2 is not None else x.new_empty(0),
                              rs1 if rs1 is not None else x.new_empty(0), rs2 if rs2 is not None else x.new_empty(0),
                              bias_t if bias_t is not None else x.new_empty(0))
        return x2

    @staticmethod
    def backward(ctx, dx2):
        plan, pc = ctx.plan, ctx.pc
        x, st1, ln1, qkv, o, lse, y1, x1, st2, ln2, h, a, y2, rs1, rs2, bias_t = ctx.saved_tensors
        rs1 = rs1 if rs1.numel() else None
        rs2 = rs2 if rs2.numel() else None
        bias_t = bias_t if ctx.has_bias else None
        M, D = x.shape
        dev = x.device
        H = pc.H
        Fdim = h.shape[1]
        dx2 = dx2.contiguous()
        g1, g2 = plan.gamma1, plan.gamma2
        folded = ctx.folded
        if folded:
            y1 = y2 = None
            flat = plan.ranges[0][2].projw._vlm_flat
            _arm_layerscale_finish(flat)
        # folded: the wgrad GEMMs and the row kernels' column sums write the RAW sums (dL/dW', dL/db'); finish_layerscale turns
        # them into the gradients of W, b and gamma once the block's last backward pass of the step has run
        wg = (lambda w: w._vlm_raw) if folded else (lambda w: w.grad)
        touch(g1, g2)
        for _, _, e in plan.ranges:
            _touch_expert(e)
        dy2 = torch.empty(M, D, device=dev, dtype=BF16)
        dy1 = torch.empty(M, D, device=dev, dtype=BF16)
        dh = torch.empty(M, Fdim, device=dev, dtype=BF16)
        dln = torch.empty(M, D, device=dev, dtype=BF16)
        dx1 = torch.empty(M, D, device=dev, dtype=F32)
        # column partials (dgamma / dbeta / dbias) of the block's four row kernels are folded by ONE launch at the end
        fold = ops.FoldBatch(dev, D) if _DEFER_FOLD else None
        grouped = _use_grouped(plan.ranges)
        if fold is not None and _EXPERTS["enabled"] and len(plan.ranges) > 1 and not grouped:
            fold.multi_stream = True  # the experts' row kernels run on two streams: fold only after the join
        # ---- FFN branch, then the attention branch up to the attention core: one independent chain per expert ----
        do = torch.empty(M, D, device=dev, dtype=BF16)
        fuse_b1 = _FUSE_FC1_BIAS and fold is not None
        if grouped:
            # the same chain with every dgrad as ONE grouped launch over the experts' row ranges; row kernels and wgrads
            # stay per expert (their parameters and gradient targets differ)
            rg = plan.ranges
            act_bwd = L.ACT_MUL_AUX if _SAVE_DERIV else L.ACT_GELU_BWD
            for r0, r1, e in rg:
                rr = slice(r0, r1)
                if folded:
                    ops.layerscale_bwd(dx2[rr], None, None, rs2[rr] if rs2 is not None else None, dy2[rr], None, _rb(e.fc2b), fold=fold)
                else:
                    ops.layerscale_bwd(dx2[rr], y2[rr], g2, rs2[rr] if rs2 is not None else None, dy2[rr], g2.grad, e.fc2b.grad,
                                       fold=fold)
            ops.gemm_grouped(dy2, [(r0, r1, wT16(e.fc2w), None, e.fc1b.grad if fuse_b1 else None) for r0, r1, e in rg], dh,
                             act=act_bwd, aux=h, col_sum_fold=fold if fuse_b1 else None)
            if not fuse_b1:
                for r0, r1, e in rg:
                    ops.colsum(dh[r0:r1], e.fc1b.grad)
            with _Side(dy2, a, dh, ln2):
                ops.gemm_wgrad_grouped(dy2, a, [(r0, r1, wg(e.fc2w)) for r0, r1, e in rg])
                ops.gemm_wgrad_grouped(dh, ln2, [(r0, r1, e.fc1w.grad) for r0, r1, e in rg])
            ops.gemm_grouped(dh, [(r0, r1, wT16(e.fc1w), None, None) for r0, r1, e in rg], dln)
            for r0, r1, e in rg:
                rr = slice(r0, r1)
                _ln2_bwd_scale1(dln[rr], x1[rr], st2[rr], e, dx1[rr], dx2[rr], y1[rr] if y1 is not None else None, g1,
                                rs1[rr] if rs1 is not None else None, dy1[rr], fold)
            ops.gemm_grouped(dy1, [(r0, r1, wT16(e.projw), None, None) for r0, r1, e in rg], do)
            with _Side(dy1, o):
                ops.gemm_wgrad_grouped(dy1, o, [(r0, r1, wg(e.projw)) for r0, r1, e in rg])
        with _ExpertStreams(plan.ranges) as es:  # (grouped: nothing left for the per-expert chains -- no side stream either)
            for idx, (r0, r1, e) in enumerate(plan.ranges if not grouped else ()):
                rr = slice(r0, r1)
                with es.on(idx):
                    if folded:
                        ops.layerscale_bwd(dx2[rr], None, None, rs2[rr] if rs2 is not None else None, dy2[rr], None, _rb(e.fc2b),
                                           fold=fold)
                    else:
                        ops.layerscale_bwd(dx2[rr], y2[rr], g2, rs2[rr] if rs2 is not None else None, dy2[rr], g2.grad,
                                           e.fc2b.grad, fold=fold)
                    # fc1 bias gradient = column sums of dh: per-tile sums from the epilogue that produces dh, folded with
                    # the block's other column partials (no atomics, no second pass over dh); without a fold batch: colsum
                    _dgrad(dy2[rr], e.fc2w, dh[rr], act=L.ACT_MUL_AUX if _SAVE_DERIV else L.ACT_GELU_BWD, aux=h[rr],
                           col_sum=e.fc1b.grad if fuse_b1 else None, col_sum_fold=fold if fuse_b1 else None)
                    if not fuse_b1:
                        ops.colsum(dh[rr], e.fc1b.grad)
                    with _Side(dy2, a, dh, ln2):
                        ops.gemm(dy2[rr], a[rr], wg(e.fc2w), ta=True, tb=True, accumulate=True)
                        ops.gemm(dh[rr], ln2[rr], e.fc1w.grad, ta=True, tb=True, accumulate=True)
                    _dgrad(dh[rr], e.fc1w, dln[rr])
                    _ln2_bwd_scale1(dln[rr], x1[rr], st2[rr], e, dx1[rr], dx2[rr], y1[rr] if y1 is not None else None, g1,
                                    rs1[rr] if rs1 is not None else None, dy1[rr], fold)
                    _dgrad(dy1[rr], e.projw, do[rr])
                    with _Side(dy1, o):
                        ops.gemm(dy1[rr], o[rr], wg(e.projw), ta=True, tb=True, accumulate=True)
        dqkv = torch.empty(M, 3 * D, device=dev, dtype=BF16)
        dln1 = torch.empty(M, D, device=dev, dtype=BF16)
        rp = pc.relpos
        # q_bias / v_bias gradients: column sums of dQ / dV per segment, taken inside the attention backward when every
        # segment lies inside ONE expert's row range (always true for the layouts Block.plan() builds)
        qb_grads, vb_grads, fused_qv = _segment_bias_grads(plan.ranges, pc.seq) if _FUSE_BIAS_GRADS else (None, None, False)
        ops.attention_bwd(qkv, o, do, lse, dqkv, pc.seq, H, bias_t=bias_t, head_row0=plan.layer * H,
                          rel_index=rp.index if rp is not None else None,
                          rel_index_t=rp.index_t if rp is not None else None, keep0=pc.keep0, keep1=pc.keep1,
                          mode=plan.mode, dbias_t=rp.holder.get("dbias_t") if rp is not None else None,
                          dq_colsum=qb_grads, dv_colsum=vb_grads,
                          bias_dense=rp.dense_for(pc.seq, plan.mode) if rp is not None else None)
        dx = torch.empty(M, D, device=dev, dtype=F32)
        if grouped:
            for r0, r1, e in plan.ranges:
                rr = slice(r0, r1)
                if e.qb is not None and not fused_qv:
                    ops.colsum(dqkv[rr, :D], e.qb.grad)
                    ops.colsum(dqkv[rr, 2 * D:], e.vb.grad)
            with _Side(dqkv, ln1):
                ops.gemm_wgrad_grouped(dqkv, ln1, [(r0, r1, e.qkvw.grad) for r0, r1, e in plan.ranges])
            ops.gemm_grouped(dqkv, [(r0, r1, wT16(e.qkvw), None, None) for r0, r1, e in plan.ranges], dln1)
            for r0, r1, e in plan.ranges:
                rr = slice(r0, r1)
                ops.layernorm_bwd(dln1[rr], x[rr], st1[rr], e.n1w, dx[rr], dres=dx1[rr], dgamma=e.n1w.grad,
                                  dbeta=e.n1b.grad, fold=fold)
        with _ExpertStreams(plan.ranges) as es:
            for idx, (r0, r1, e) in enumerate(plan.ranges if not grouped else ()):
                rr = slice(r0, r1)
                with es.on(idx):
                    if e.qb is not None and not fused_qv:
                        ops.colsum(dqkv[rr, :D], e.qb.grad)
                        ops.colsum(dqkv[rr, 2 * D:], e.vb.grad)
                    with _Side(dqkv, ln1):
                        ops.gemm(dqkv[rr], ln1[rr], e.qkvw.grad, ta=True, tb=True, accumulate=True)
                    _dgrad(dqkv[rr], e.qkvw, dln1[rr])
                    ops.layernorm_bwd(dln1[rr], x[rr], st1[rr], e.n1w, dx[rr], dres=dx1[rr], dgamma=e.n1w.grad,
                                      dbeta=e.n1b.grad, fold=fold)
        if fold is not None:
            fold.flush()
        if folded:
            flat.ls_pending.add(plan.layer)
        if ctx.hook is not None:
            ctx.hook(plan.layer)
        dbias = None
        if bias_t is not None and ctx.needs_input_grad[1] and rp is not None and not rp.holder.get("routed"):
            # the real gradient sits in relpos.holder["dbias_t"]; autograd only needs ONE defined tensor per table handle to
            # reach _TableT.backward (a zero from every block evaluation cost 22 [144, R] additions per step)
            rp.holder["routed"] = True
            dbias = _zero_scalar(dev).expand_as(bias_t)
        return dx, dbias, None, None, None, None


def run_block(x, plan: BlockPlan, pc: PassCtx, training: bool, hook=None):
    bias_t = pc.relpos.bias_t if pc.relpos is not None else None
    return _BlockFn.apply(x, bias_t, plan, pc, training, hook)


# ----------------------------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) through the MFMA GEMM; bf16 in/out (fp32 out for act = "tanh"), wgrad accumulated into W.grad in place.
    Covers heads.py (Pooler.dense + tanh, ITMHead.fc, IFMHead.fc, MLMHead.transform.dense + GELU / decoder).  x may be a row-strided
    view (the cls rows hidden[:, 0] of a [B, N, D] tensor): the GEMMs take its leading dimension, no gather copy."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype != BF16:
            x2 = x2.to(BF16)
        if x2.stride(1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]):
            x2 = x2.contiguous()
        M, K = x2.shape
        N = weight.shape[0]
        Np = (N + 63) // 64 * 64
        gelu = act == "gelu"
        buf = torch.empty(M, Np, device=x.device, dtype=BF16)
        pre = torch.empty(M, Np, device=x.device, dtype=BF16) if gelu else None
        ops.gemm(x2, w16(weight), buf[:, :N], bias=bias, act=L.ACT_GELU if gelu else L.ACT_NONE,
                 aux=pre[:, :N] if gelu else None)
        y = ops.tanh_fwd(buf[:, :N]) if act == "tanh" else None
        ctx.save_for_backward(x2, pre if gelu else (y if y is not None else x2.new_empty(0)))
        ctx.weight, ctx.bias, ctx.act, ctx.shape, ctx.N, ctx.Np = weight, bias, act, x.shape, N, Np
        return (y if y is not None else buf[:, :N]).view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, gy):
        x2, saved = ctx.saved_tensors
        weight, bias, N, Np, act = ctx.weight, ctx.bias, ctx.N, ctx.Np, ctx.act
        M, K = x2.shape
        gy2 = gy.reshape(M, N)
        if gy2.dtype not in (BF16, F32):
            gy2 = gy2.float()
        if gy2.stride(1) != 1:
            gy2 = gy2.contiguous()
        if (act is None and gy2.dtype == BF16 and gy2.stride() == (Np, 1) and gy2.storage_offset() == 0
                and _is_padded_grad(gy2, M, Np)):
            # the fused cross-entropy's gradient: already the zero-padded bf16 [M, Np] operand (registered by _CrossEntropyFn)
            dy = torch.as_strided(gy2, (M, Np), (Np, 1))
        elif act is None and Np == N and gy2.dtype == BF16:
            dy = gy2  # nothing to pad or cast: the gradient itself is the GEMMs' operand
        else:
            # ONE launch: upstream gradient x activation derivative -> the zero-padded bf16 operand (csrc/frontops.hip)
            mode = {"gelu": ops.ACT_BWD_GELU, "tanh": ops.ACT_BWD_TANH, None: ops.ACT_BWD_NONE}[act]
            dy = ops.act_bwd(gy2, saved[:, :N] if act is not None else None, mode, Np)
        touch(weight if weight.requires_grad else None, bias if bias is not None and bias.requires_grad else None)
        if bias is not None and bias.requires_grad:
            if N % 8 == 0:
                ops.colsum(dy[:, :N], bias.grad)
            elif N <= 64:
                ops.colsum_small(dy[:, :N], bias.grad)
            elif bias.grad.is_contiguous():
                # ragged vocabulary (30 522 = 8 * 3 815 + 2): the vector kernel over the whole 8-column groups, the tail apart
                N8 = N // 8 * 8
                ops.colsum(dy[:, :N8], bias.grad[:N8])
                ops.colsum_small(dy[:, N8:N], bias.grad[N8:])
            else:
                bias.grad.add_(dy[:, :N].float().sum(0))
        if weight.requires_grad:
            ops.gemm(dy[:, :N], x2, weight.grad, ta=True, tb=True, accumulate=True)
        dx = None
        if ctx.needs_input_grad[0]:
            # ragged N (vocab 30522): the reduction runs over the zero-padded Np columns of dy; the weight rows
            # past N are whatever follows in the flat buffer (finite), multiplied by those zeros
            wpad = w16(weight) if Np == N else torch.as_strided(w16(weight), (Np, K), (K, 1))
            if Np >= 8192 and ((M + 127) // 128) * ((K + 127) // 128) <= 128:
                # a long reduction into a small output (the MLM decoder: 42 tiles of 128 x 128 for 256 CUs, 475 us): the
                # library's split-K path accumulates fp32 slices, one round of workgroups instead of a sixth of one
                dx32 = torch.zeros(M, K, device=x2.device, dtype=F32)
                ops.gemm(dy, wpad, dx32, tb=True, accumulate=True)
                dx = dx32.to(BF16)
            else:
                dx = torch.empty(M, K, device=x2.device, dtype=BF16)
                ops.gemm(dy, wpad, dx, tb=True)
            dx = dx.view(ctx.shape)
        return dx, None, None, None


def linear(x, weight, bias=None, gelu=False, act=None):
    """act: None, "gelu" (bf16 out, `gelu=True` is the older spelling) or "tanh" (fp32 out)."""
    return _LinearFn.apply(x, weight, bias, "gelu" if gelu else act)


class _L2NormFn(torch.autograd.Function):
    """y = x / ||x||_2 over the last dim in fp32 (the contrastive heads' feature normalisation, objectives.py:248-300): ONE HIP
    kernel forward and one backward -- dx = (g - y (g . y)) / ||x||, written in x's dtype (round 6; round 4's form took two torch
    kernels forward, four backward, and autograd cast the fp32 gradient back to bf16 in another)."""

    @staticmethod
    def forward(ctx, x):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        y, inv = ops.l2norm_fwd(x2)
        ctx.save_for_backward(y, inv)
        ctx.dtype, ctx.shape = x.dtype, x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, g):
        y, inv = ctx.saved_tensors
        g2 = g.reshape(y.shape)
        if g2.dtype != F32 or not g2.is_contiguous():
            g2 = g2.float().contiguous()
        return ops.l2norm_bwd(g2, y, inv, ctx.dtype).view(ctx.shape)


_FUSED_LOSS = os.environ.get("VLM_FUSED_LOSS", "1") != "0"  # A/B switch: 0 = torch's cross_entropy / x / x.norm() graphs


def l2_normalize(x):
    if not _FUSED_LOSS or not x.is_cuda or x.dtype not in (BF16, F32):
        x = x.float()
        return x / x.norm(dim=-1, keepdim=True)
    return _L2NormFn.apply(x)


class _ContrastiveFn(torch.autograd.Function):
    """(loss, logits_per_image, exp(log_scale)) of the symmetric contrastive cross-entropy on normalised features, forward AND
    gradient in two launches (csrc/lossops.hip vlm_contrastive; reference objectives.py:274-300 / :393-445):
    all_* = this rank's B rows first, then the other ranks' (no gradient through those, as in the reference).  The loss is a
    scalar: its upstream gradient scales the stored feature / scale gradients in ONE launch (vlm_scale_by_scalar)."""

    @staticmethod
    def forward(ctx, img, txt, log_scale, others_img, others_txt):
        B = img.shape[0]
        all_img = img if others_img is None else torch.cat([img, others_img])
        all_txt = txt if others_txt is None else torch.cat([txt, others_txt])
        ls = log_scale.detach().reshape(1).float()
        out3, logits, d_img, d_txt = ops.contrastive(all_img.contiguous(), all_txt.contiguous(), B, ls)
        ctx.save_for_backward(out3, d_img, d_txt)
        ctx.ls_shape = log_scale.shape
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)  # an unused exp(log_scale) output arrives as None, not as a zero to multiply and add
        return out3[0], logits, out3[2]

    @staticmethod
    def backward(ctx, g_loss, _g_logits, g_scale):
        out3, d_img, d_txt = ctx.saved_tensors
        if g_loss is None:
            g_loss = torch.zeros((), device=out3.device, dtype=F32)
        gi, gt, gs = ops.scale_by_scalar([d_img, d_txt, out3[1:2]], g_loss.reshape(1).float())
        gs = gs.view(ctx.ls_shape)
        if g_scale is not None:  # somebody differentiates the returned scale itself (d exp(l) / d l = exp(l)): not on the hot path
            gs = gs + (g_scale * out3[2]).view(ctx.ls_shape)
        return gi, gt, gs, None, None


def contrastive_loss(img, txt, log_scale, others_img=None, others_txt=None):
    """loss, logits_per_image [n, n], exp(log_scale): img / txt = this rank's normalised fp32 features [B, D]."""
    return _ContrastiveFn.apply(img, txt, log_scale, others_img, others_txt)


_WVEC = {}


class _WeightedSumFn(torch.autograd.Function):
    """sum_k w_k * loss_k of scalar device losses in ONE launch (training_step's `sum(losses)`, compute_ifm's
    `(ifm_weight * a + b) * 0.5`: torch issues a multiply / add per term forward and a multiply per term backward)."""

    @staticmethod
    def forward(ctx, weights, *terms):
        ctx.weights, ctx.dev = weights, terms[0].device
        return ops.weighted_sum([t.detach().reshape(1) for t in terms], weights)

    @staticmethod
    def backward(ctx, g):
        w = ctx.weights
        if all(x == 1.0 for x in w):
            return (None,) + (g,) * len(w)
        key = (w, str(ctx.dev))
        wv = _WVEC.get(key)
        if wv is None:
            if len(_WVEC) > 64:
                _WVEC.clear()
            wv = _WVEC[key] = torch.tensor(w, dtype=F32).to(ctx.dev)
        gw = g * wv  # one launch for all terms
        return (None,) + tuple(gw[k] for k in range(len(w)))


def weighted_sum(terms, weights=None):
    """sum_k weights[k] * terms[k] for scalar losses (weights default to 1); anything that is not a handful of fp32 CUDA scalars
    goes through torch."""
    terms = list(terms)
    weights = tuple(float(w) for w in (weights if weights is not None else [1.0] * len(terms)))
    if (_FUSED_LOSS and 1 <= len(terms) <= 8 and all(torch.is_tensor(t) and t.is_cuda and t.dtype == F32 and t.numel() == 1 for t in terms)):
        return _WeightedSumFn.apply(weights, *terms)
    out = 0
    for t, w in zip(terms, weights):
        out = out + (t if w == 1.0 else t * w)
    return out


class _SmallCrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits [rows, V], labels) for a handful of classes (the ITM head's [3B, 2]): loss and gradient in one launch."""

    @staticmethod
    def forward(ctx, logits, labels):
        loss, d = ops.small_cross_entropy(logits, labels)
        ctx.save_for_backward(d)
        ctx.dtype = logits.dtype
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        (gd,) = ops.scale_by_scalar([d], g.reshape(1).float())
        return (gd if ctx.dtype == F32 else gd.to(ctx.dtype)), None


def small_cross_entropy(logits, labels):
    if (_FUSED_LOSS and logits.is_cuda and logits.dim() == 2 and logits.dtype in (BF16, F32) and logits.stride(1) == 1
            and logits.shape[1] <= 64 and labels.dtype == torch.int64):
        return _SmallCrossEntropyFn.apply(logits, labels.contiguous())
    return torch.nn.functional.cross_entropy(logits.float(), labels)


_PADDED_GRADS = {}  # data_ptr -> (weakref to the buffer, rows, padded columns) of gradient buffers born zero-padded (consumed once by _LinearFn.backward)


def _is_padded_grad(g, M, Np):
    ent = _PADDED_GRADS.pop(g.data_ptr(), None)
    return ent is not None and ent[0]() is not None and ent[1:] == (M, Np)


class _CrossEntropyFn(torch.autograd.Function):
    """mean-reduced F.cross_entropy(logits, labels, ignore_index) on bf16 logits [rows, V] as they leave the decoder GEMM
    (objectives.py:88-143), through the two row kernels of csrc/lossops.hip: no fp32 copy of the [880, 30 522] logits, and the
    gradient is born as the padded bf16 matrix the decoder's dgrad / wgrad GEMMs read (_LinearFn.backward takes it as is)."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        loss_rows, lse = ops.cross_entropy_fwd(logits, labels, ignore_index)
        # the kernels' own validity rule (a label outside [0, V) gets neither loss nor gradient, where torch raises a device
        # assert): such rows must not sit in the denominator either.  One launch: (mean over the counted rows, 1 / count)
        out2 = ops.cross_entropy_reduce(loss_rows, labels, logits.shape[1], ignore_index)
        ctx.save_for_backward(logits, labels, lse, out2)
        ctx.ignore_index = ignore_index
        return out2[0]  # no counted row: 0 / 0 = nan, like F.cross_entropy

    @staticmethod
    def backward(ctx, g):
        logits, labels, lse, out2 = ctx.saved_tensors
        (scale,) = ops.scale_by_scalar([out2[1:2]], g.reshape(1).float())  # upstream gradient / count, on the device
        d = ops.cross_entropy_bwd(logits, labels, lse, scale, ctx.ignore_index)
        if len(_PADDED_GRADS) > 64:
            _PADDED_GRADS.clear()  # entries nobody consumed (a caller that is not _LinearFn): never grow
        # the marker is tied to the padded buffer OBJECT: while it lives its memory cannot be handed to another tensor, and once
        # it is gone (the gradient was re-materialised by autograd) a recycled pointer finds a dead reference, not a match
        _PADDED_GRADS[d.data_ptr()] = (weakref.ref(d._base), d.shape[0], d._base.shape[1])
        return d, None, None


def cross_entropy(logits, labels, ignore_index=-100):
    """F.cross_entropy(logits.float(), labels, ignore_index=ignore_index) for 2-D bf16 CUDA logits whose rows are 16-B aligned (the
    MLM head's); anything else goes to torch."""
    if (_FUSED_LOSS and logits.is_cuda and logits.dtype == BF16 and logits.dim() == 2 and logits.stride(1) == 1 and logits.stride(0) % 8 == 0
            and logits.data_ptr() % 16 == 0 and labels.dtype == torch.int64):
        return _CrossEntropyFn.apply(logits, labels.contiguous(), ignore_index)
    return torch.nn.functional.cross_entropy(logits.float(), labels, ignore_index=ignore_index)


class _LayerNormFn(torch.autograd.Function):
    """LayerNorm over the last dim of a fp32 [rows, D] matrix, bf16 or fp32 out (transformer.norm, MLM transform)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, out_f32):
        shape = x.shape
        x2 = x.reshape(-1, shape[-1]).float().contiguous()
        M, D = x2.shape
        y = torch.empty(M, D, device=x.device, dtype=F32 if out_f32 else BF16)
        st = torch.empty(M, 2, device=x.device, dtype=F32)
        ops.layernorm_fwd(x2, weight, bias, eps, y, st)
        ctx.save_for_backward(x2, st)
        ctx.weight, ctx.bias, ctx.shape, ctx.in_dtype = weight, bias, shape, x.dtype
        return y.view(shape)

    @staticmethod
    def backward(ctx, gy):
        x2, st = ctx.saved_tensors
        M, D = x2.shape
        g2 = gy.reshape(M, D).contiguous()
        if g2.dtype not in (BF16, F32):
            g2 = g2.float()
        dx = torch.empty(M, D, device=x2.device, dtype=F32)
        touch(ctx.weight if ctx.weight.requires_grad else None, ctx.bias if ctx.bias.requires_grad else None)
        ops.layernorm_bwd(g2, x2, st, ctx.weight, dx, dgamma=ctx.weight.grad if ctx.weight.requires_grad else None,
                          dbeta=ctx.bias.grad if ctx.bias.requires_grad else None)
        return dx.view(ctx.shape).to(ctx.in_dtype), None, None, None, None


def layer_norm(x, weight, bias, eps, out_f32=False):
    return _LayerNormFn.apply(x, weight, bias, eps, out_f32)


def _rows2d(g, D):
    g2 = g.reshape(-1, D)
    if g2.dtype not in (BF16, F32):
        g2 = g2.float()
    return g2 if g2.stride(1) == 1 else g2.contiguous()


class _FeatureViewsFn(torch.autograd.Function):
    """The views of a pass's final feature matrix x [B T + B I, D] (segment-major) that its result exposes -- text_feats [B, T, D],
    image_feats [B, I, D], and the cls rows of both as [B, D] views (reference vilt_module.py:1136-1156, :1199-1223, :1346-1375:
    `x[:, :T]`, `x[:, T:]`, `[:, 0]`) -- as ONE autograd node: the gradient of x is assembled from whichever views received one in a
    single launch (ops.scatter_rows), where autograd materialises a zero tensor + a strided copy per view and adds them up."""

    @staticmethod
    def forward(ctx, x, B, T, I):
        ctx.set_materialize_grads(False)
        D = x.shape[1]
        nt = B * T
        ctx.geom = (x.shape[0], D, B, T, I, x.dtype)
        text = x[:nt].view(B, T, D)
        image = x[nt:].view(B, I, D)
        tcls = torch.as_strided(x, (B, D), (T * D, 1), x.storage_offset()) if T else x[:0]
        icls = torch.as_strided(x, (B, D), (I * D, 1), x.storage_offset() + nt * D) if I else x[:0]
        return text, image, tcls, icls

    @staticmethod
    def backward(ctx, g_text, g_image, g_tcls, g_icls):
        R, D, B, T, I, dtype = ctx.geom
        nt = B * T
        src = []
        if g_text is not None and T:
            src.append((_rows2d(g_text, D), 0, 1))
        if g_image is not None and I:
            src.append((_rows2d(g_image, D), nt, 1))
        if g_tcls is not None and T:
            src.append((_rows2d(g_tcls, D), 0, T))
        if g_icls is not None and I:
            src.append((_rows2d(g_icls, D), nt, I))
        dev = src[0][0].device if src else None
        if not src:
            return None, None, None, None
        return ops.scatter_rows(R, D, dtype, src, dev), None, None, None


def feature_views(x, B, T, I):
    """(text_feats [B, T, D], image_feats [B, I, D], text cls rows [B, D], image cls rows [B, D]) of x [B T + B I, D]: see
    _FeatureViewsFn.  Falls back to plain views where the kernel does not apply (CPU, a width that is not a multiple of 4)."""
    D = x.shape[1]
    nt = B * T
    if x.is_cuda and x.is_contiguous() and x.dtype in (BF16, F32) and D % 4 == 0 and torch.is_grad_enabled() and x.requires_grad:
        return _FeatureViewsFn.apply(x, B, T, I)
    text, image = x[:nt].view(B, T, D), x[nt:].view(B, I, D)
    return text, image, text[:, 0], image[:, 0]


class _RowRangeFn(torch.autograd.Function):
    """t[a:b] along the first dim as one autograd node whose backward is ONE launch (zero fill and copy together)."""

    @staticmethod
    def forward(ctx, t, a, b):
        ctx.shape, ctx.dtype, ctx.a = t.shape, t.dtype, a
        return t[a:b]

    @staticmethod
    def backward(ctx, g):
        D = ctx.shape[-1]
        inner = 1
        for d in ctx.shape[1:-1]:
            inner *= d
        g2 = _rows2d(g, D)
        return ops.scatter_rows(ctx.shape[0] * inner, D, ctx.dtype, [(g2, ctx.a * inner, 1)], g2.device).view(ctx.shape), None, None


def row_range(t, a, b):
    """t[a:b] (first dim); on the GPU with a one-launch backward (objectives.py: `infer["text_feats"][:bsz]`, `cls_feats[bsz:]`)."""
    if (t.is_cuda and t.dim() >= 2 and t.is_contiguous() and t.dtype in (BF16, F32) and t.shape[-1] % 4 == 0 and torch.is_grad_enabled()
            and t.requires_grad):
        return _RowRangeFn.apply(t, a, t.shape[0] if b is None else b)
    return t[a:b]


class _EmbeddingFn(torch.autograd.Function):
    """nn.Embedding gather (BertEmbeddings.word_embeddings, vilt_module.py:63) whose backward adds the token rows straight
    into the flat gradient buffer instead of torch's sort + dense [vocab, D] gradient + accumulate."""

    @staticmethod
    def forward(ctx, ids, weight, padding_idx):
        ctx.save_for_backward(ids)
        ctx.weight, ctx.padding_idx = weight, padding_idx
        return torch.nn.functional.embedding(ids, weight.detach())

    @staticmethod
    def backward(ctx, gy):
        (ids,) = ctx.saved_tensors
        w = ctx.weight
        if w.requires_grad:
            touch(w)
            g2 = gy.reshape(-1, gy.shape[-1])
            if g2.dtype != F32 or g2.stride(1) != 1:
                g2 = g2.float().contiguous()
            ops.embedding_bwd(g2, ids.reshape(-1).contiguous(), w.grad, -1 if ctx.padding_idx is None else ctx.padding_idx)
        return None, None, None


def embedding(ids, weight, padding_idx=None):
    if getattr(weight, "_vlm_flat", None) is None or not weight.is_cuda:
        return torch.nn.functional.embedding(ids, weight, padding_idx=padding_idx)  # not flattened (CPU construction)
    return _EmbeddingFn.apply(ids, weight, padding_idx)


class _PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed (vision_transformer.py:714-728) as im2col + MFMA GEMM; row 0 of every image is left for the cls
    token (the GEMM writes `bias` there).  The image is a leaf: no dX."""

    @staticmethod
    def forward(ctx, image, weight, bias, patch):
        B, C, Hh, Ww = image.shape
        Dm = weight.shape[0]
        rows = 1 + (Hh // patch) * (Ww // patch)
        K = C * patch * patch
        cols = torch.empty(B * rows, K, device=image.device, dtype=BF16)
        ops.patch_im2col(image.contiguous().float(), cols, patch, 1)
        out = torch.empty(B * rows, Dm, device=image.device, dtype=F32)
        ops.gemm(cols, w16(weight).view(Dm, K), out, bias=bias)
        ctx.save_for_backward(cols)
        ctx.weight, ctx.bias, ctx.B, ctx.rows = weight, bias, B, rows
        return out.view(B, rows, Dm)

    @staticmethod
    def backward(ctx, gy):
        (cols,) = ctx.saved_tensors
        weight, bias = ctx.weight, ctx.bias
        Dm = weight.shape[0]
        g = gy.reshape(-1, Dm)
        g16 = torch.empty(g.shape, device=g.device, dtype=BF16)
        g16.copy_(g)
        # rows 0 of every image carry no patch (zero im2col row): they add nothing to dW; their bias share is
        # excluded because the reference's conv never produced that row
        gv = g16.view(ctx.B, ctx.rows, Dm)
        gv[:, 0].zero_()
        touch(weight if weight.requires_grad else None, bias if bias is not None and bias.requires_grad else None)
        if weight.requires_grad:
            ops.gemm(g16, cols, weight.grad.view(Dm, -1), ta=True, tb=True, accumulate=True)
        if bias is not None and bias.requires_grad:
            ops.colsum(g16, bias.grad)
        return None, None, None, None


def patch_embed(image, weight, bias, patch):
    return _PatchEmbedFn.apply(image, weight, bias, patch)


class TextSpec:
    """What _PassRowsFn needs to produce the text rows of a pass itself (BertEmbeddings.forward + the modality type row 0,
    reference vilt_module.py:51-63, :1111-1113): ids [B, T] int64, the word table, BERT's own token-type table (row 0 is added
    before the LayerNorm), the LayerNorm, and the dropout as (u fp32 [B * T, D] or None, p, scale): an element is kept where
    u >= p and multiplied by scale."""
    __slots__ = ("ids", "word", "padding_idx", "bert_type", "gamma", "beta", "eps", "u", "p", "scale")

    def __init__(self, ids, word, padding_idx, bert_type, gamma, beta, eps, u=None, p=0.0, scale=1.0):
        self.ids, self.word, self.padding_idx, self.bert_type = ids, word, padding_idx, bert_type
        self.gamma, self.beta, self.eps, self.u, self.p, self.scale = gamma, beta, eps, u, p, scale


def _g(p):
    return p.grad if (p is not None and p.requires_grad) else None


class _PassRowsFn(torch.autograd.Function):
    """The token matrix a pass starts from, x = [text rows ; image rows] (segment-major, fp32), both halves produced in place:
      text rows (`text`: a TextSpec)  dropout(LayerNorm(word[ids] + bert_type[0])) + token_type[0] -- ONE launch (csrc/frontops.hip;
                                      torch: gather, add, LayerNorm, dropout, add, copy into x), or `trows` copied in;
      image rows                      visual_embed fused into the patch-embed GEMM (reference
                                      vision_transformer.py:952-991 + vilt_module.py:1111-1117):
        image row (b, 0)     = cls_token + token_type[idx]
        image row (b, 1 + p) = conv(patch p) + conv_bias + token_type[idx]
      The GEMM writes its rows straight into x with (conv_bias + token_type[idx]) as its bias; the B lead rows are then overwritten.
    Backward writes every parameter gradient straight into the flat gradient buffer (touch()): the word rows by atomics, the
    LayerNorm / type-row / conv-bias / cls gradients as column sums folded in a fixed order, the conv weight through the wgrad GEMM.
    Only `trows` (when the caller made the text rows itself) gets a gradient through autograd."""

    @staticmethod
    def forward(ctx, trows, image, weight, bias, cls_token, tt_weight, tt_idx, patch, text):
        Dm = tt_weight.shape[1]
        dev = tt_weight.device
        B = rows = K = 0
        if image is not None:
            B, C, Hh, Ww = image.shape
            rows = 1 + (Hh // patch) * (Ww // patch)
            K = C * patch * patch
        nt = text.ids.numel() if text is not None else (0 if trows is None else trows.shape[0])
        x = torch.empty(nt + B * rows, Dm, device=dev, dtype=F32)
        stats = None
        if text is not None:
            stats = ops.text_rows_fwd(text.ids.reshape(-1), text.word.detach(), text.bert_type.detach()[0], text.gamma.detach(),
                                      text.beta.detach(), text.eps, x[:nt], text.u, text.p, text.scale, add1=tt_weight.detach()[0])
        elif nt:
            x[:nt].copy_(trows)
        cols = None
        if image is not None:
            cols = torch.empty(B * rows, K, device=dev, dtype=BF16)
            im = image if (image.dtype == F32 and image.is_contiguous()) else image.contiguous().float()
            ops.patch_im2col(im, cols, patch, 1)
            pre = ops.image_rows_prep(bias.detach() if bias is not None else None, tt_weight.detach()[tt_idx], cls_token.detach())
            ops.gemm(cols, w16(weight).view(Dm, K), x[nt:], bias=pre[0])
            ops.image_lead_rows(x[nt:], B, rows, pre[1])
        ctx.save_for_backward(*[t for t in (cols, stats) if t is not None])
        ctx.has = (cols is not None, stats is not None)
        ctx.weight, ctx.bias, ctx.cls, ctx.tt, ctx.text = weight, bias, cls_token, tt_weight, text
        ctx.geom = (B, rows, Dm, nt, tt_idx)
        return x

    @staticmethod
    def backward(ctx, gx):
        saved = list(ctx.saved_tensors)
        cols = saved.pop(0) if ctx.has[0] else None
        stats = saved.pop(0) if ctx.has[1] else None
        weight, bias, cls, tt, text = ctx.weight, ctx.bias, ctx.cls, ctx.tt, ctx.text
        B, rows, Dm, nt, tt_idx = ctx.geom
        if gx.dtype != F32 or gx.stride(1) != 1:
            gx = gx.float().contiguous()
        if cols is not None:
            touch(*[p for p in (weight, bias, cls, tt) if p is not None and p.requires_grad])
            g16 = ops.image_rows_bwd(gx[nt:], B, rows, _g(bias), _g(tt)[tt_idx] if _g(tt) is not None else None, _g(cls))
            if weight.requires_grad:
                ops.gemm(g16, cols, weight.grad.view(Dm, -1), ta=True, tb=True, accumulate=True)
        if text is not None:
            touch(*[p for p in (text.word, text.bert_type, text.gamma, text.beta, tt) if p.requires_grad])
            gb, gt = _g(text.bert_type), _g(tt)
            ops.text_rows_bwd(gx[:nt], text.ids.reshape(-1), text.word.detach(), text.bert_type.detach()[0], text.gamma.detach(), stats,
                              text.u, text.p, text.scale, _g(text.word), text.padding_idx, gt[0] if gt is not None else None,
                              _g(text.beta), _g(text.gamma), gb[0] if gb is not None else None)
        return (gx[:nt] if (text is None and nt and ctx.needs_input_grad[0]) else None), None, None, None, None, None, None, None, None


def pass_rows(trows, image, weight, bias, cls_token, tt_weight, tt_idx, patch, text=None):
    """x = [text rows ; image rows of `image`] for a pass: the text rows either given (`trows`) or made here from `text` (a
    TextSpec); `image` may be None (a text-only pass from a TextSpec); see _PassRowsFn."""
    if text is not None and trows is not None:
        raise ValueError("pass_rows: text rows OR a TextSpec")
    return _PassRowsFn.apply(trows, image, weight, bias, cls_token, tt_weight, tt_idx, patch, text)

