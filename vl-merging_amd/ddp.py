"""Data-parallel gradient exchange for the flat gradient buffer: one process per GPU, RCCL all-reduce over xGMI.

The reference gets this from pytorch_lightning's DDP wrapper (run.py:263-288).  Here a bucket is a
contiguous slice of engine.FlatParams.flat_g: {heads}, {block 11} ... {block 0}, {embeddings + bias table}.  A block's
slice is final once every pass that used the block has run its backward (weights are shared by up to six passes per
step); the fused block function reports each backward through `on_block_backward`, and the slice's all-reduce is
launched on a side stream as soon as the count is complete, overlapping the remaining backward.  Parameters that get
no gradient in a step (position_embeddings, mask_token, ...) stay zero in the flat buffer: the
reducer needs no unused-parameter discovery (the optimizer keeps the structural set of parameters a backward pass has
written, engine.FlatParams.touched, and skips the rest as HF AdamW skips p.grad is None).  The 1/world average is folded into the fused AdamW kernel (grad_scale).
"""
import os
import re

import torch
import torch.distributed as dist

from . import engine

_BLOCK = re.compile(r"transformer\.blocks\.(\d+)\.")


class _Works:
    """Several asynchronous collectives of one bucket behind a single wait()."""

    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()


class _WireHandle(_Works):
    """A bucket reduced in a narrower wire dtype: wait(), then widen back into the fp32 gradient slice (on the waiting
    stream: one small cast launch per bucket)."""

    def __init__(self, works, wire, grad):
        super().__init__(works)
        self.wire, self.grad = wire, grad

    def wait(self):
        super().wait()
        self.grad.copy_(self.wire)


class FlatGradReducer:
    def __init__(self, model, process_group=None, force_collectives=False, sharded=False, comm_dtype=None,
                 collective="allreduce", standin=None, cu_budget="auto"):
        """force_collectives: issue the all-reduces even at world size 1 (a single-GPU smoke test of the RCCL path).
        comm_dtype: None / torch.float32 -> the fp32 gradients travel as they are (what Lightning's DDP does with the
        reference's fp32 master gradients); torch.bfloat16 -> every bucket is cast into a bf16 wire buffer, reduced there
        and widened back (half the xGMI bytes: 272 / 471 MB per step instead of 545 / 942; the SUM then rounds to 8 bits
        per addition, so this is an option, not the default).
        collective: "allreduce" (default) or "rs_ag": reduce_scatter_tensor into the rank's own 1/W chunk followed by
        all_gather_into_tensor of the bucket -- the two halves of a ring all-reduce as separate RCCL calls, each of which
        can use the direct xGMI links; same result, meaningful only with the nccl backend.
        sharded: the reference's `ddp_sharded` plugin (run.py:231-232, fairscale OSS + ShardedDDP) on the flat buffers:
        every bucket is REDUCE-SCATTERED instead of all-reduced (rank r receives the sum of its 1/W chunk of the
        bucket), FusedAdamW keeps m / v only for those chunks and updates only them, and the updated fp32 parameters are
        ALL-GATHERED bucket by bucket after the step.  Same bytes on the wire as one all-reduce (reduce-scatter +
        all-gather is how RCCL's ring all-reduce is built), 1/W of the optimizer state and of the AdamW traffic."""
        # standin (measurement tool, world size 1 only): dict(cus=k, lds_kb=.., gbps=..) -- every bucket's
        # collective is REPLACED by what it would cost this GPU's other streams: k workgroups that hold a CU each for the time
        # the bucket's bytes need at `gbps` per GPU (vlm_debug_occupy), and a device copy of the bucket (the HBM traffic of the
        # local reduction), both on the communication stream behind the same waits a real collective has.
        self.standin = dict(standin) if standin else None
        self._standin_buf = None
        self.force = force_collectives or bool(self.standin)
        self.sharded = bool(sharded)
        self.comm_dtype = comm_dtype if comm_dtype in (torch.bfloat16,) else None
        if collective not in ("allreduce", "rs_ag"):
            raise ValueError("collective must be 'allreduce' or 'rs_ag'")
        self.collective = collective
        self._wire = None  # bf16 wire buffer, same offsets as the flat gradient buffer (allocated on first use)
        self.model = model
        self.flat = model._flat
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        # Data-parallel default from the single-GPU contention stand-in (profiles/r05_contention.json, tools/contention_sweep.py:
        # every bucket's collective replaced by k workgroups holding a CU each + a copy of the bucket): the GEMM grids plan for 248
        # of the 256 CUs (neutral at k <= 16, 70.6 -> 69.2 ms at k = 32).  cu_budget: "auto" = that default, and ONLY where it
        # can matter -- gradients on a GPU exchanged by RCCL's own kernels (backend nccl), world size > 1, no VLM_GEMM_CUS in the
        # environment; an integer = that budget; None = leave the library alone.  The budget is PROCESS-global (it sizes every GEMM
        # grid and split-K slice count of this process, so a world-size-1 run and a world-size-N run differ in summation
        # order): `cu_budget_set` says what this reducer did, `close()` puts the previous value back, bench.py reports the value.
        self.cu_budget_set, self._cu_budget_prev = None, None
        want = None
        if cu_budget == "auto":
            nccl = dist.is_initialized() and dist.get_backend(process_group) == "nccl"
            if self.world > 1 and nccl and self.flat.flat_g.is_cuda and not os.environ.get("VLM_GEMM_CUS"):
                want = 248
        elif cu_budget is not None:
            want = int(cu_budget)
        if want is not None:
            from . import _lib as L
            lib = L.get_lib()
            if lib.vlm_device_cus() >= 256 or cu_budget != "auto":
                self._cu_budget_prev = lib.vlm_device_cus()
                lib.vlm_set_cu_budget(want)
                self.cu_budget_set = want
        names_by_block = {}
        early, late = [], []
        for n in self.flat.names:
            m = _BLOCK.search(n)
            if m:
                names_by_block.setdefault(int(m.group(1)), []).append(n)
            elif n.startswith(("text_embeddings", "token_type_embeddings", "transformer.patch_embed",
                               "transformer.cls_token", "transformer.mask_token", "relative_position_bias_table",
                               "temporal_relative_position_bias_table")):
                early.append(n)
            else:
                late.append(n)
        self.block_slices = {i: self.flat.slice_of(ns) for i, ns in names_by_block.items()}
        # heads (everything downstream of the last block in every pass) are final once the last block's last backward
        # has run: their slice goes out with that block's, under the rest of backward; embeddings come at the very end
        self.late_slices = [self.flat.slice_of(late)] if late else []
        self.tail_slices = [self.flat.slice_of(early)] if early else []
        self.last_layer = max(names_by_block) if names_by_block else None
        self.expected = {i: 0 for i in self.block_slices}
        self.seen = {i: 0 for i in self.block_slices}
        self.handles = []
        self.comm_stream = torch.cuda.Stream() if self.flat.flat_g.is_cuda else None
        self.counting = True
        model._grad_hook = self.on_block_backward
        self.measure = False      # bench.py turns this on for the timed steps
        self.defer_tail = False   # True: finish_backward leaves the last (embeddings) all-reduce in flight, see wait_tail
        self.accumulate = False   # True: this backward is a non-final micro-batch, keep the gradients local
        self._attached = None
        self._tail_handles = []
        self._wait_events = []
        self.timeline_on = False  # bench.py: per-bucket issue / complete times of the LAST step (bucket_timeline())
        self._t0, self._timeline = None, []

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def attach(self, optimizer, defer_tail=True):
        """Wire an optimizer to this reducer: the 1/world gradient average, and (defer_tail) the embeddings' slice whose
        all-reduce stays in flight while AdamW updates everything else.  The reducer owns both halves of that contract,
        so a caller cannot enable the deferral without the wait (bench.py and run.py both go through here)."""
        optimizer.grad_scale = self.grad_scale
        self.defer_tail = bool(defer_tail) and (self.world > 1 or self.force)
        tr = self.tail_range()
        optimizer.tail_sync = (tr[0], tr[1], self.wait_tail) if tr is not None else None
        if self.sharded and self.world > 1:
            optimizer.set_shard(self.own_ranges(), self.gather_params)
        self._attached = optimizer
        return self

    def close(self):
        """Put the process-global CU budget back to what it was before this reducer changed it (no-op otherwise)."""
        if self.cu_budget_set is not None:
            from . import _lib as L
            lib = L.get_lib()
            raw = self._cu_budget_prev
            lib.vlm_set_cu_budget(0)
            if raw is not None and raw != lib.vlm_device_cus():
                lib.vlm_set_cu_budget(raw)  # the previous value was itself a budget (VLM_GEMM_CUS / an outer reducer)
            self.cu_budget_set = None

    def begin_step(self):
        # a deferred tail all-reduce nobody waited for (optimizer step skipped, or defer_tail set by hand without
        # tail_sync) must land before the gradient buffer is written again
        if self._tail_handles:
            self.wait_tail()
        if self.timeline_on and self.comm_stream is not None:
            self._t0 = torch.cuda.Event(enable_timing=True)
            self._t0.record()
            self._timeline = []
        for i in self.seen:
            self.seen[i] = 0
        self.handles = []

    def note_forward(self, layer):
        """Called by the first (counting) step for every block evaluation to learn the static use counts."""
        self.expected[layer] += 1

    # ---- sharded mode: chunk r of bucket [lo, hi) belongs to rank r ---------------------------------------------------
    def buckets(self):
        """All buckets (flat ranges) in flat-buffer order: tail (embeddings), blocks, late (heads)."""
        return sorted(list(self.tail_slices) + list(self.block_slices.values()) + list(self.late_slices))

    def bucket_plan(self):
        """What travels per step, for the bench line: the buckets in the order their collectives are issued during backward
        (heads with the last block's, blocks 11 .. 0, the embeddings' slice last and left in flight under AdamW), each with its
        size on the wire -- so the first multi-GPU measurement can be read against the link model."""
        wire = 2 if self.comm_dtype is not None else 4
        order = []
        for i in sorted(self.block_slices, reverse=True):
            lo, hi = self.block_slices[i]
            order.append({"what": "block %d" % i, "MB": round((hi - lo) * wire / 1e6, 2)})
            if i == self.last_layer:
                for lo2, hi2 in self.late_slices:
                    order.append({"what": "heads", "MB": round((hi2 - lo2) * wire / 1e6, 2)})
        for lo, hi in self.tail_slices:
            order.append({"what": "embeddings (deferred under AdamW)" if self.defer_tail else "embeddings",
                          "MB": round((hi - lo) * wire / 1e6, 2)})
        return {"count": len(order), "total_MB": round(sum(b["MB"] for b in order), 2), "in_issue_order": order}

    def bucket_timeline(self):
        """Per bucket of the last step (timeline_on): milliseconds from begin_step to the collective's issue on the
        communication stream and to its completion -- the first multi-GPU run reads its exposed time off this (synchronises)."""
        if not self._timeline or self._t0 is None:
            return None
        torch.cuda.synchronize()
        names = {}
        for i, (lo, hi) in self.block_slices.items():
            names[(lo, hi)] = "block %d" % i
        for lo, hi in self.late_slices:
            names[(lo, hi)] = "heads"
        for lo, hi in self.tail_slices:
            names[(lo, hi)] = "embeddings"
        wire = 2 if self.comm_dtype is not None else 4
        return [{"what": names.get((lo, hi), "[%d,%d)" % (lo, hi)), "MB": round((hi - lo) * wire / 1e6, 2),
                 "issue_ms": round(self._t0.elapsed_time(e0), 3), "done_ms": round(self._t0.elapsed_time(e1), 3)}
                for lo, hi, e0, e1 in self._timeline]

    def own_chunk(self, lo, hi, rank=None):
        n = hi - lo
        if n % self.world:
            raise RuntimeError("sharded mode: bucket of %d elements does not split over %d ranks (buckets are multiples "
                               "of 64 elements: use a world size that divides 64)" % (n, self.world))
        c = n // self.world
        r = self.rank if rank is None else rank
        return lo + r * c, lo + (r + 1) * c

    def own_ranges(self):
        return [self.own_chunk(lo, hi) for lo, hi in self.buckets()]

    def _reduce(self, lo, hi):
        """The gradient collective of one bucket (async): all-reduce, or reduce-scatter into the rank's own chunk.
        Returns an object with .wait() that makes the CURRENT stream see the reduced fp32 gradients of [lo, hi)."""
        if self.standin is not None:
            return self._standin_reduce(lo, hi)
        nccl = dist.get_backend(self.group) == "nccl"
        if self.comm_dtype is not None and not (self.sharded and self.world > 1):
            if self._wire is None:
                self._wire = torch.empty(self.flat.flat_g.shape, device=self.flat.flat_g.device, dtype=self.comm_dtype)
            wire = self._wire[lo:hi]
            wire.copy_(self.flat.flat_g[lo:hi])  # on the communication stream, behind the bucket's last wgrad
            works = self._collect(wire, lo, hi, nccl)
            return _WireHandle(works, wire, self.flat.flat_g[lo:hi])
        buf = self.flat.flat_g[lo:hi]
        if self.sharded and self.world > 1:
            clo, chi = self.own_chunk(lo, hi)
            if nccl:
                return dist.reduce_scatter_tensor(self.flat.flat_g[clo:chi], buf, group=self.group, async_op=True)
            # gloo (CPU tests, several ranks on one device) has no reduce-scatter: the all-reduce leaves the same sum
            # in the own chunk (the other chunks are ignored by the sharded optimizer)
            return dist.all_reduce(buf, group=self.group, async_op=True)
        return _Works(self._collect(buf, lo, hi, nccl))

    def _standin_reduce(self, lo, hi):
        from . import ops
        st = self.standin
        nbytes = (hi - lo) * 4
        # a ring all-reduce over W ranks moves 2 (W - 1) / W of the bucket per GPU; W = 8 assumed
        usec = int(2.0 * 7.0 / 8.0 * nbytes / (float(st.get("gbps", 300.0)) * 1e3)) + 1
        if self._standin_buf is None or self._standin_buf.numel() < hi - lo:
            self._standin_buf = torch.empty(max(hi - lo for lo, hi in self.buckets()), device=self.flat.flat_g.device,
                                            dtype=torch.float32)
        ops.debug_occupy(int(st.get("cus", 16)), 512, int(st.get("lds_kb", 64)) * 1024, usec)
        self._standin_buf[:hi - lo].copy_(self.flat.flat_g[lo:hi])
        ev = torch.cuda.Event()
        ev.record()

        class _Ev:
            def wait(self_inner):
                torch.cuda.current_stream().wait_event(ev)
        return _Ev()

    def _collect(self, buf, lo, hi, nccl):
        """all-reduce of `buf` (the bucket [lo, hi) in its wire dtype) as one or two RCCL calls."""
        if self.collective == "rs_ag" and nccl and (hi - lo) % self.world == 0:
            c = (hi - lo) // self.world
            mine = buf[self.rank * c:(self.rank + 1) * c]
            w1 = dist.reduce_scatter_tensor(mine, buf, group=self.group, async_op=True)
            w2 = dist.all_gather_into_tensor(buf, mine, group=self.group, async_op=True)  # same RCCL stream: ordered behind w1
            return [w1, w2]
        return [dist.all_reduce(buf, group=self.group, async_op=True)]

    def gather_params(self):
        """Sharded mode, after the optimizer step: every bucket's fp32 parameters from their owners (in place)."""
        if not self.sharded or self.world == 1:
            return
        works = []
        for lo, hi in self.buckets():
            clo, chi = self.own_chunk(lo, hi)
            full, mine = self.flat.flat_p[lo:hi], self.flat.flat_p[clo:chi]
            if dist.get_backend(self.group) == "nccl":
                works.append(dist.all_gather_into_tensor(full, mine, group=self.group, async_op=True))
            else:
                c = chi - clo
                works.append(dist.all_gather([full[r * c:(r + 1) * c] for r in range(self.world)], mine.clone(),
                                             group=self.group, async_op=True))
        for w in works:
            w.wait()

    def _launch(self, lo, hi):
        if self.world == 1 and not self.force:
            return
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            ws = engine.wgrad_stream()
            if ws is not None:
                self.comm_stream.wait_stream(ws)  # the slice's weight gradients are produced on the side stream
            with torch.cuda.stream(self.comm_stream):
                if self.timeline_on and self._t0 is not None:
                    # issue -> complete of this bucket on the communication stream (which then waits for the collective itself:
                    # collectives of one process group are serial anyway; the compute stream is not involved)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    h = self._reduce(lo, hi)
                    h.wait()
                    e1.record()
                    self._timeline.append((lo, hi, e0, e1))
                    self.handles.append(h)
                else:
                    self.handles.append(self._reduce(lo, hi))
        else:
            self.handles.append(self._reduce(lo, hi))

    def on_block_backward(self, layer):
        self.seen[layer] += 1
        if self.counting or self.accumulate:
            return
        if self.seen[layer] == self.expected[layer]:
            # folded LayerScale (engine.FlatParams.finish_layerscale): the block's raw sums become the gradients of W, b and
            # gamma BEFORE its slice travels (the transform is linear, but a sharded rank only ever sees its own chunk of the sum)
            self.flat.finish_layerscale(layer)
            self._launch(*self.block_slices[layer])
            if layer == self.last_layer:
                for lo, hi in self.late_slices:
                    self._launch(lo, hi)

    def finish_backward(self):
        """After loss.backward(): reduce what is left, then make the compute stream wait for all buckets.
        With `accumulate` set (a non-final micro-batch of gradient accumulation, run.py grad_steps) nothing is sent: the
        flat buffer keeps summing local gradients and the final micro-batch reduces the sum once."""
        if self.accumulate:
            if self.counting:
                self.expected = dict(self.seen)
                self.counting = False
            return
        self.flat.finish_layerscale()  # nothing pending in a normal step (end-of-backward callback / per-bucket finish)
        if self.counting:
            # first step: use counts were unknown during backward -> reduce every block now, remember the counts
            self.expected = dict(self.seen)
            self.counting = False
            for i in sorted(self.block_slices, reverse=True):
                self._launch(*self.block_slices[i])
            for lo, hi in self.late_slices:
                self._launch(lo, hi)
        else:
            for i, n in self.seen.items():
                if n != self.expected[i]:
                    raise RuntimeError("block %d ran %d backward passes, expected %d: static use counts changed"
                                       % (i, n, self.expected[i]))
        n_before_tail = len(self.handles)
        for lo, hi in self.tail_slices:
            self._launch(lo, hi)
        self._tail_handles = self.handles[n_before_tail:] if self.defer_tail else []
        self._wait(self.handles[:n_before_tail] if self.defer_tail else self.handles)

    def _wait(self, handles):
        timed = self.comm_stream is not None and self.measure and (self.world > 1 or self.force)
        if timed:  # how long the compute stream sits waiting for the buckets = the exposed (non-overlapped) part
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        for h in handles:
            h.wait()  # the current (compute) stream waits for this collective only
        if timed:
            e1.record()
            self._wait_events.append((e0, e1))

    def tail_range(self):
        """Flat range [lo, hi) of the slice whose all-reduce finish_backward leaves in flight (defer_tail), or None."""
        if not self.defer_tail or not self.tail_slices:
            return None
        return min(lo for lo, _ in self.tail_slices), max(hi for _, hi in self.tail_slices)

    def wait_tail(self):
        """Make the compute stream wait for the embeddings' all-reduce: called by the optimizer after it has updated
        everything else (FusedAdamW.tail_sync), so ~1 ms of AdamW runs under the last collective."""
        self._wait(self._tail_handles)
        self._tail_handles = []

    def exposed_wait_ms(self):
        """Total milliseconds the compute stream waited for gradient all-reduces since the last call (synchronises)."""
        if not self._wait_events:
            return 0.0
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._wait_events)
        self._wait_events = []
        return ms
