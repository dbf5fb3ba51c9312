#!/usr/bin/env python
"""bench.py -- VL samples/sec (fwd+bwd+optimizer) of the VLMo hot path on MI355X, plus the merge kernel's GB/s.

  python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: this process starts the N ranks itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): base_vl `ufo`, 384x384 patch 16, text length 40, per-GPU batch 22,
task_mlm_itm_ifm (one training_step = mlm joint pass + ifm image/text passes + 3 itm joint passes, backward, fused
AdamW step), train mode (DropPath / dropout live), synthetic batch, random-init weights, bf16 MFMA
GEMMs with fp32 accumulation and an fp32 residual stream.  N > 1: weak scaling, gradient all-reduce over RCCL.
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

FLOP_PER_SAMPLE_384 = 1835.1e9  # fwd 611.7 GFLOP x 3
MFMA_PEAK_TFLOPS = 2500.0       # MI355X dense bf16 (MI355X_MICROARCH.md)
HBM_PEAK_GBPS = 8000.0
F64_MFMA_PEAK_TFLOPS = 78.6     # the Gram / RegMean leg's roofline (v_mfma_f64_16x16x4_f64)


_T0 = time.time()


def _phase(msg):
    """Progress on stderr (the ONE JSON line owns stdout): which leg the run is in and how long it has been going."""
    sys.stderr.write("[bench %7.1f s] %s\n" % (time.time() - _T0, msg))
    sys.stderr.flush()


def synthetic_batch(*a, **kw):
    """The synthetic batch of SURVEY.md 8(d): lives in the package (vl_merging_amd.synthetic) so that run.py does not import the
    benchmark; kept here as the name the tests and tools call."""
    ge.import_package()
    return importlib.import_module("vl_merging_amd.synthetic").synthetic_batch(*a, **kw)


class GemmTimer:
    """HIP-event timing of every GEMM launch on the launching (= torch current) stream during the timed region."""

    def __init__(self, ops):
        self.ops = ops
        self.orig = ops.gemm
        self.records = []
        self.on = False

    def _bracket(self, fn, flop):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        self.records.append((e0, e1, flop))
        return r

    def install(self):
        orig, orig_g, orig_w = self.orig, self.ops.gemm_grouped, self.ops.gemm_wgrad_grouped

        def timed(a, b, out, ta=False, tb=False, **kw):
            if not self.on:
                return orig(a, b, out, ta, tb, **kw)
            M, N = out.shape
            K = a.shape[0] if ta else a.shape[1]
            return self._bracket(lambda: orig(a, b, out, ta, tb, **kw), 2.0 * M * N * K)

        def timed_grouped(a, groups, out, **kw):  # all_moe: one launch over the experts' row ranges
            if not self.on:
                return orig_g(a, groups, out, **kw)
            rows = sum(r1 - r0 for r0, r1, *_ in groups)
            return self._bracket(lambda: orig_g(a, groups, out, **kw), 2.0 * rows * out.shape[1] * a.shape[1])

        def timed_wgrad(dy, x, groups, **kw):
            if not self.on:
                return orig_w(dy, x, groups, **kw)
            rows = sum(r1 - r0 for r0, r1, _ in groups)
            return self._bracket(lambda: orig_w(dy, x, groups, **kw), 2.0 * rows * dy.shape[1] * x.shape[1])

        self.ops.gemm = timed
        self.ops.gemm_grouped = timed_grouped
        self.ops.gemm_wgrad_grouped = timed_wgrad

    def summary(self):
        if not self.records:
            return None
        torch.cuda.synchronize()
        t = sum(e0.elapsed_time(e1) for e0, e1, _ in self.records) * 1e-3
        fl = sum(f for _, _, f in self.records)
        n = len(self.records)
        return {"launches": n, "seconds": t, "flop": fl, "avg_us": t / n * 1e6, "tflops": fl / t / 1e12}


GEMM_KERNELS = ("vlm_gemm_big_kernel", "vlm_gemm_bigT_kernel", "vlm_gemm_kernel")  # one vlm_gemm_bf16 call runs one of them
GEMM_HELPERS = ("splitk_reduce_kernel",)  # second launch of a wgrad call: its bytes count, its launches do not


ATTN_KERNELS = {"fwd": ("attn_fwd2_kernel",), "bwd": ("attn_bwd_dq2_kernel", "attn_bwd_dkvb_kernel", "attn_dbias_fold_kernel")}
# what a default pretrain step launches from the attention / GEMM families: a committed PMC file that lacks one of them was
# made from another build of the kernels and must not vouch for this run
EXPECTED_IN_PROFILE = GEMM_KERNELS + GEMM_HELPERS + ATTN_KERNELS["fwd"] + ATTN_KERNELS["bwd"]


def _git_blob_sha1(data):
    import hashlib
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_traffic(kernels, helpers=(), with_meta=False):
    """HBM-side bytes per launch of a kernel (or, launch-weighted, of a family of kernels that serve the same call) from the
    committed rocprofv3 --pmc passes over this same bench command (the NEWEST profiles/r*_pmc_traffic.json, made by
    tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled on gfx950).
    Counters cannot be read from inside the timed process, so the figure is only as good as the file: the meta record names the
    file and its git blob hash, and a file whose kernel list lacks a kernel this build launches (EXPECTED_IN_PROFILE) is
    REFUSED (value None, the reason in meta["stale"], a line on stderr) -- tests/test_bench_gpu.py fails on that."""
    if isinstance(kernels, str):
        kernels = (kernels,)
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(here, "profiles", "r*_pmc_traffic.json")), reverse=True)
    value, meta = None, {"file": None}
    if files:
        path = files[0]
        try:
            with open(path, "rb") as f:
                raw = f.read()
            doc = json.loads(raw.decode())
            ks = doc["kernels"]
            meta = {"file": "profiles/" + os.path.basename(path), "git_blob": _git_blob_sha1(raw), "kernels_in_file": len(ks)}
            if "total_bytes_per_step" in doc:
                meta["total_bytes_per_step"] = doc["total_bytes_per_step"]
            missing = [k for k in EXPECTED_IN_PROFILE if k not in ks]
            if missing:
                meta["stale"] = "the file has no record of %s: it was not made from this build" % ", ".join(missing)
                sys.stderr.write("[bench] roofline.traffic REFUSED: %s (%s)\n" % (meta["stale"], meta["file"]))
            else:
                found = [ks[k] for k in kernels if k in ks]
                if found:
                    total = sum(k["hbm_bytes_per_launch"] * k["launches"] for k in found + [ks[h] for h in helpers if h in ks])
                    value = total / sum(k["launches"] for k in found)
        except (OSError, KeyError, ValueError) as e:
            meta = {"file": "profiles/" + os.path.basename(path), "stale": "unreadable: %r" % (e,)}
    return (value, meta) if with_meta else value


class AttnTimer:
    """HIP-event timing of every fused-attention call (forward: one launch; backward: dQ + fused dK/dV/bias-gradient + fold)
    on the launching stream, in the same bracketed steps as GemmTimer.  flop = the ALGORITHMIC count: 4 n_q n_k 64 per
    (sample, head) forward (Q K^T and P V), twice that backward -- the convention of profiles/r0*_pmc_mfma_busy.json."""

    def __init__(self, ops, gemm_timer):
        self.ops, self.gt = ops, gemm_timer
        self.rec = {"fwd": [], "bwd": []}
        self.L = importlib.import_module("vl_merging_amd._lib")

    def _flop(self, seq, H, mode):
        n0, n1 = seq.n0, seq.n1
        pairs = (n0 * n0 + n1 * n1) if mode == self.L.ATTN_SEPARATE else (n0 + n1) ** 2
        return 4.0 * 64 * H * seq.B * pairs

    def install(self):
        of, ob = self.ops.attention_fwd, self.ops.attention_bwd

        def bracket(kind, fn, flop):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn()
            e1.record()
            self.rec[kind].append((e0, e1, flop))
            return r

        def fwd(qkv, out, lse, seq, H, **kw):
            if not self.gt.on:
                return of(qkv, out, lse, seq, H, **kw)
            return bracket("fwd", lambda: of(qkv, out, lse, seq, H, **kw), self._flop(seq, H, kw.get("mode", self.L.ATTN_JOINT)))

        def bwd(qkv, out, dout, lse, dqkv, seq, H, **kw):
            if not self.gt.on:
                return ob(qkv, out, dout, lse, dqkv, seq, H, **kw)
            return bracket("bwd", lambda: ob(qkv, out, dout, lse, dqkv, seq, H, **kw),
                           2.0 * self._flop(seq, H, kw.get("mode", self.L.ATTN_JOINT)))

        self.ops.attention_fwd, self.ops.attention_bwd = fwd, bwd

    def summary(self):
        out = {}
        for kind, recs in self.rec.items():
            if not recs:
                continue
            torch.cuda.synchronize()
            t = sum(e0.elapsed_time(e1) for e0, e1, _ in recs) * 1e-3
            fl = sum(f for _, _, f in recs)
            out[kind] = {"kernel": " + ".join(ATTN_KERNELS[kind]), "bound": "mfma", "achieved": fl / t / 1e12,
                         "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": fl / t / 1e12 / MFMA_PEAK_TFLOPS,
                         "launches": len(recs), "avg_call_us": t / len(recs) * 1e6, "flop_convention": "algorithmic: 4 n_q n_k 64 per (sample, head) forward, 2 x backward"}
        return out or None


def calibrate(dev):
    """What this box attains with vendor code (report nominal AND attainable peaks): a large bf16 GEMM
    through torch (hipBLASLt) and a device-to-device copy.  Context for roofline.frac; never used as its denominator."""
    def ev_time(fn, n):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / n
    a = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
    t_mm = ev_time(lambda: torch.matmul(a, b), 10)
    src = torch.empty(1 << 28, device=dev, dtype=torch.float32)  # 1 GiB
    dst = torch.empty_like(src)
    t_cp = ev_time(lambda: dst.copy_(src), 10)
    return {"hipblaslt_bf16_8192_tflops": 2.0 * 8192 ** 3 / t_mm / 1e12,
            "d2d_copy_GBps": 2.0 * src.numel() * 4 / t_cp / 1e9}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_merge_baseline():
    """The merge oracle (oracle/merge_oracle.py: C restatement of the reference's arithmetic, numpy float64 for RegMean)
    on this host, base-size all_moe blocks: interpolation and task-vector over all 12 layers (median of 5 after 1
    warm-up, GB/s over the same algorithmic bytes as the GPU figure), RegMean on a bounded 2-layer sample."""
    import numpy as np
    from oracle import merge_oracle as mo
    from oracle import synth
    from oracle.detweights import det_array, det_gram
    D, Fd = 768, 3072
    shapes = synth.block_shapes(D, Fd, "all_moe")
    sd = {k: det_array(k, s) for k, (s, dt) in shapes.items()}
    cfg = dict(vlffn_start_layer_index=10, only_activate_used_experts=False, merge_ratio=0.5, sum_lambda=0.75,
               scaling_for_non_diag=0.9, loss_names={"irtr": 1})
    central = {k: det_array(k, s, 7) for k, (s, dt) in synth.block_shapes(D, Fd, "ufo").items()}

    def med(fn, n=5):
        fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[n // 2], min(ts), max(ts)

    out = {}
    t, lo, hi = med(lambda: mo.merge_weights(sd, cfg))
    moved = 0
    res = mo.merge_weights(sd, cfg)
    for k, v in res.items():  # algorithmic bytes: 4 B written + 4 B per source read, for every merged output
        srcs = [kk for kk in sd if kk != k and kk.replace(".v.", ".").replace(".l.", ".").replace(".vl.", ".") == k]
        if srcs:
            moved += np.asarray(v).nbytes * (1 + len(srcs))
    out["interpolation"] = {"seconds_median": t, "seconds_min_max": [lo, hi], "GBps": moved / t / 1e9, "bytes": moved}
    t, lo, hi = med(lambda: mo.sum_task_vectors(sd, cfg, central))
    out["task_vector"] = {"seconds_median": t, "seconds_min_max": [lo, hi]}
    layers = (0, 11)
    sub = {k: v for k, v in sd.items() if int(k.split(".")[2]) in layers}
    for k, (shp, dt) in synth.block_shapes(D, Fd, "ufo").items():
        if int(k.split(".")[2]) not in layers and "gamma" not in k:
            sub[k] = central[k]
    grams = {k: det_gram(k, shp[0]) for k, shp in synth.gram_shapes(D, Fd).items() if int(k.split(".")[2]) in layers}
    t0 = time.perf_counter()
    mo.regmean(sub, cfg, grams)
    t = time.perf_counter() - t0
    out["regmean"] = {"seconds": t, "sample": "2 of 12 layers (8 weight matrices: 6 inverses of 768^2, 2 of 3072^2), numpy float64",
                      "seconds_scaled_to_12_layers": 6 * t}
    return out


def cpu_baseline():
    """The oracle (parity-pinned CPU restatement of the reference) on this host's cores: base_vl ufo, 224x224, B=2,
    mlm+itm+ifm fwd+bwd (BASELINE configs[0]) at the best of a short thread-count sweep, median and spread; CPU model and
    thread count stated; plus the merge oracle (cpu_merge_baseline).  The `_`-prefixed entries (the oracle's loss, weights
    and batch) feed parity_gates() and are dropped from the JSON line."""
    from oracle import vlmo_ref as R
    pkg_vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    torch.manual_seed(0)
    idx, nrel, _, allrel = pkg_vm.build_relative_position_indices((14, 14), 40, 196, 40)
    D, Fd, V = 768, 3072, 30522
    g = torch.Generator().manual_seed(0)

    def rn(*s, std=0.02):
        return (torch.randn(*s, generator=g) * std).requires_grad_(True)

    sd = {"relative_position_bias_table": rn(allrel, 144), "logit_scale": torch.tensor(2.659).requires_grad_(True),
          "logit_vl_scale": torch.tensor(2.659).requires_grad_(True),
          "text_embeddings.word_embeddings.weight": rn(V, D), "text_embeddings.token_type_embeddings.weight": rn(2, D),
          "text_embeddings.LayerNorm.weight": torch.ones(D, requires_grad=True),
          "text_embeddings.LayerNorm.bias": torch.zeros(D, requires_grad=True), "token_type_embeddings.weight": rn(2, D),
          "transformer.cls_token": rn(1, 1, D), "transformer.patch_embed.proj.weight": rn(D, 3, 16, 16),
          "transformer.patch_embed.proj.bias": torch.zeros(D, requires_grad=True),
          "transformer.norm.weight": torch.ones(D, requires_grad=True),
          "transformer.norm.bias": torch.zeros(D, requires_grad=True), "pooler.dense.weight": rn(D, D),
          "pooler.dense.bias": torch.zeros(D, requires_grad=True), "mlm_score.transform.dense.weight": rn(D, D),
          "mlm_score.transform.dense.bias": torch.zeros(D, requires_grad=True),
          "mlm_score.transform.LayerNorm.weight": torch.ones(D, requires_grad=True),
          "mlm_score.transform.LayerNorm.bias": torch.zeros(D, requires_grad=True),
          "mlm_score.decoder.weight": rn(V, D), "mlm_score.bias": torch.zeros(V, requires_grad=True),
          "itm_score.fc.weight": rn(2, D), "itm_score.fc.bias": torch.zeros(2, requires_grad=True)}
    for n in ("ifm_text_proj", "ifm_image_proj", "ifm_vl_text_proj", "ifm_vl_image_proj"):
        sd[n + ".fc.weight"] = rn(D, D)
    for i in range(12):
        p = f"transformer.blocks.{i}."
        sd[p + "gamma_1"] = torch.full((D,), 0.1, requires_grad=True)
        sd[p + "gamma_2"] = torch.full((D,), 0.1, requires_grad=True)
        sd[p + "attn.q_bias"] = torch.zeros(D, requires_grad=True)
        sd[p + "attn.v_bias"] = torch.zeros(D, requires_grad=True)
        sd[p + "attn.qkv.weight"] = rn(3 * D, D)
        sd[p + "attn.proj.weight"] = rn(D, D)
        sd[p + "attn.proj.bias"] = torch.zeros(D, requires_grad=True)
        sd[p + "mlp.fc1.weight"] = rn(Fd, D)
        sd[p + "mlp.fc1.bias"] = torch.zeros(Fd, requires_grad=True)
        sd[p + "mlp.fc2.weight"] = rn(D, Fd)
        sd[p + "mlp.fc2.bias"] = torch.zeros(D, requires_grad=True)
        for nm in ("norm1", "norm2"):
            sd[p + nm + ".weight"] = torch.ones(D, requires_grad=True)
            sd[p + nm + ".bias"] = torch.zeros(D, requires_grad=True)
    a = R.Arch("ufo")
    b = synthetic_batch(2, 224, 40, V, 1234, "cpu")["vl"]
    batch = dict(b)
    batch["image"] = b["image"][0]

    def iteration():
        t0 = time.perf_counter()
        out = R.pretrain_step(sd, a, idx, batch)
        out["total_loss"].backward()
        for v in sd.values():
            v.grad = None
        return time.perf_counter() - t0, float(out["total_loss"].detach())

    # Thread count: torch's default (every logical CPU) oversubscribes a B = 2 problem -- on the pool's 256-thread hosts ONE
    # iteration at the default took longer than the whole rest of the run (round 5: 1.6 s at 16 threads, 2.7 at 32, 5.3 at 64,
    # > 800 s at 256), and the figure moved +-60 % between boxes in round 4.  Ascending sweep after a common warm-up, stopped
    # at the first count that is slower than the one before; then the median of four iterations at the best count.
    ncpu = os.cpu_count() or 1
    cands = [c for c in (4, 8, 16, 32, 64) if c <= ncpu] or [ncpu]
    torch.set_num_threads(cands[0])
    _phase("cpu baseline: warm-up iteration at %d threads (candidates %r)" % (cands[0], cands))
    _, oracle_loss = iteration()  # warm-up (allocator, oneDNN primitives); its loss is the parity gate's reference value
    sweep = {}
    for c in cands:
        torch.set_num_threads(c)
        sweep[c], _ = iteration()
        _phase("cpu baseline: %d threads %.1f s per iteration" % (c, sweep[c]))
        if len(sweep) > 1 and sweep[c] > min(v for k, v in sweep.items() if k != c):
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    times = [sweep[best]] + [iteration()[0] for _ in range(3)]
    per_iter = sorted(times)[len(times) // 2]
    res = {"value": 2.0 / per_iter, "unit": "samples/s", "cores": best, "kind": "port",
           "cpu_model": _cpu_model(), "logical_cpus": ncpu,
           "seconds_per_iteration": {"median": per_iter, "min": min(times), "max": max(times)},
           "thread_sweep_seconds": {str(c): sweep[c] for c in sweep},
           "sample": "oracle/vlmo_ref.py pretrain_step fwd+bwd, base_vl ufo 224^2 T=40 B=2 (BASELINE configs[0]), fp32: one "
                     "warm-up, one iteration per thread count in %r (ascending, stopped at the first slower one), then the "
                     "median of 4 at the best count (%d threads)" % (sorted(sweep), best),
           "_oracle_loss": oracle_loss, "_sd": sd, "_batch": b}
    _phase("cpu baseline: merge oracle")
    try:
        try:  # numpy's BLAS pool (RegMean's float64 inverse) is as oversubscribed by default as torch's was
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=max(best, 16)):
                res["merge"] = cpu_merge_baseline()
        except ImportError:
            res["merge"] = cpu_merge_baseline()
    except Exception as e:  # the baseline must never take the GPU number down with it
        res["merge"] = {"error": repr(e)}
    return res


TASKS = {
    # task -> (named configs in front of the arch, default per-GPU batch, algorithmic FLOP per sample at 384^2,
    #          BASELINE configs index per arch, workload text)
    "pretrain": (("task_mlm_itm_ifm_square_randaug_base_vl", "step200k"), 22, 1835.1e9, {"ufo": 1, "all_moe": 2},
                 "task_mlm_itm_ifm"),
    "irtr": (("task_finetune_irtr_coco_square_randaug_base_image384",), 20, 353.5e9, {"ufo": 4, "all_moe": 4},
             "task_finetune_irtr_coco"),
}


def _standin_env(world):
    """VLM_DDP_STANDIN="cus[:lds_kb[:gbps]]" (world size 1 only): every gradient bucket's collective is replaced by its
    single-GPU stand-in -- k workgroups holding a CU each + a copy of the bucket on the communication stream
    (ddp.FlatGradReducer(standin=...), tools/contention_sweep.py -> profiles/r05_contention.json)."""
    v = os.environ.get("VLM_DDP_STANDIN", "")
    if not v or world > 1:
        return None
    parts = v.split(":")
    st = {"cus": int(parts[0])}
    if len(parts) > 1:
        st["lds_kb"] = int(parts[1])
    if len(parts) > 2:
        st["gbps"] = float(parts[2])
    return st


SETUP_STEPS = int(os.environ.get("VLM_BENCH_SETUP_STEPS", "3"))


class TrainLeg:
    """One data-parallel training workload of BASELINE.json built the way run.py builds it: model -> engine -> fused AdamW +
    schedule -> gradient reducer (buckets overlapped with backward) -> one synthetic batch per rank.  `step()` is one
    fwd + bwd + (all-reduce) + AdamW + schedule step; every rank of the job builds the same leg and steps it together."""

    def __init__(self, mods, task, arch, B, image_size, rank, world, dev, force_dist=False):
        cfgmod, vm, vu, ddp = mods["cfg"], mods["vm"], mods["vu"], mods["ddp"]
        names, _, _, _, _ = TASKS[task]
        over = dict(image_size=image_size, vit="vit_base_patch16_%d" % image_size, per_gpu_batchsize=B, num_gpus=world)
        if task == "pretrain":
            over["vl_mlm_prob"] = 0.25
        self.cfg = cfg = cfgmod.make_config(*names, arch, **over)
        torch.manual_seed(0)
        self.model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).to(dev)
        self.model.train()
        self.model.setup_engine()
        (self.opt,), (self.sch,) = vu.set_schedule(self.model, max_steps=cfg["max_steps"] or 100000)
        # VLM_GRAD_COMM=bf16: gradients travel as bf16 (half the xGMI bytes); VLM_GRAD_COLLECTIVE=rs_ag: reduce-scatter +
        # all-gather instead of one all-reduce per bucket.  Defaults: fp32, all-reduce (what the reference's DDP does).
        self.reducer = ddp.FlatGradReducer(
            self.model, force_collectives=force_dist, sharded=os.environ.get("VLM_SHARDED", "0") != "0",  # ddp_sharded (run.py:231-232)
            comm_dtype=torch.bfloat16 if os.environ.get("VLM_GRAD_COMM", "fp32") == "bf16" else None,
            collective=os.environ.get("VLM_GRAD_COLLECTIVE", "allreduce"), standin=_standin_env(world))
        # the embeddings' all-reduce overlaps the AdamW update of everything else (same wiring as run.py)
        self.reducer.attach(self.opt, defer_tail=os.environ.get("VLM_DEFER_TAIL_ALLREDUCE", "1") != "0")
        b = synthetic_batch(B, image_size, cfg["max_text_len"], cfg["vocab_size"], 1234 + rank, dev)
        self.batch = b if cfg["tasks"] is not None else b["vl"]  # vilt_module.py:1485: {"vl": batch} only with `tasks`
        self.B, self.world, self.dist, self.rank = B, world, (world > 1 or force_dist), rank
        # Part of BUILDING the leg, before any warm-up or timed step: SETUP_STEPS full steps so that the caching allocator's pools,
        # the lazily loaded code objects and the gradient reducer's use counts exist (the first process on a fresh box showed one
        # 150-ms step among 65-ms ones as late as its fifth step: gpurun_out/r05/call_j.txt).  Disclosed in the JSON line as
        # `setup_steps`; the W warm-up steps and the K timed steps of the contract follow unchanged.
        for _ in range(SETUP_STEPS):
            self.step()
        torch.cuda.synchronize()

    def step(self):
        self.reducer.begin_step()
        loss = self.model.training_step(self.batch, 0)
        loss.backward()
        self.reducer.finish_backward()
        self.opt.step()
        self.sch["scheduler"].step()
        return loss

    def fence(self):
        if self.dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, steps, warmup, dev, per_step=None):
        """`warmup` untimed steps, then EXACTLY `steps` steps between two barrier + synchronize fences; the time is the MAX
        over the ranks.  per_step(i): hook in front of timed step i (the headline leg switches its GEMM timer there)."""
        loss = None
        for _ in range(warmup):
            loss = self.step()
        self.fence()
        # no cyclic-GC pause inside the timed region (a step allocates thousands of short-lived Python objects)
        import gc
        gc.collect()
        gc_was = gc.isenabled()
        gc.disable()
        self.reducer.measure = True
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]  # per-step durations (no sync inside the region)
        t0 = time.perf_counter()
        marks[0].record()
        for it in range(steps):
            if per_step is not None:
                per_step(it)
            # per-bucket issue / complete times of the LAST timed step only (the communication stream then waits for each of its
            # collectives in turn: harmless, but kept out of the other steps)
            self.reducer.timeline_on = self.rank == 0 and it == steps - 1
            loss = self.step()
            marks[it + 1].record()
        self.reducer.timeline_on = False
        self.fence()
        dt = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        self.step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        if self.world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        return dt, loss

    def close(self):
        self.reducer.begin_step()  # drains a deferred tail nobody waited for
        self.reducer.close()       # the process-global CU budget goes back to what it was
        torch.cuda.synchronize()
        del self.model, self.opt, self.sch, self.reducer, self.batch
        torch.cuda.empty_cache()


def secondary_train_legs(mods, dev, rank, world, headline, force_dist=False, steps=4, warm=2, image_size=384, batch=None):
    """The other data-parallel BASELINE configs, run by EVERY rank after the timed region (configs[2] all_moe B = 22 and
    configs[4] irtr on ufo B = 20 at 384^2, whichever is not the headline): whole-job samples/s over `steps` timed steps after
    `warm` warm-ups, fenced and MAX-reduced like the headline -- one `bench.py --gpus N` gives all three DP numbers."""
    out = {}
    for key, task, arch in (("ufo_b22", "pretrain", "ufo"), ("all_moe_b22", "pretrain", "all_moe"), ("irtr_ufo_b20", "irtr", "ufo")):
        if (task, arch) == headline:
            continue
        B = batch or TASKS[task][1]
        err = torch.zeros(1, device=dev)
        try:
            leg = TrainLeg(mods, task, arch, B, image_size, rank, world, dev, force_dist)
        except Exception as e:  # a secondary leg must never take the headline number down with it
            out[key] = {"error": repr(e)}
            err += 1
            leg = None
        if world > 1:  # a rank that could not build the leg must not leave its peers in the leg's collectives
            dist.all_reduce(err)
        if float(err) > 0:
            out.setdefault(key, {"error": "another rank failed to build this leg"})
            if leg is not None:
                leg.close()
            continue
        dt, _ = leg.timed(steps, warm, dev)
        out[key] = {"samples_per_s": B * world * steps / dt, "ms_per_step": dt / steps * 1e3, "batch": B, "n_gpus": world,
                    "steps": steps, "warmup": warm, "exposed_comm_ms_per_step": leg.reducer.exposed_wait_ms() / steps,
                    "workload": "BASELINE configs[%d]" % TASKS[task][3][arch]}
        leg.close()
    return out


def secondary_benches(dev):
    """configs[3]'s other legs on ONE GPU (rank 0, N = 1 only: the merge is 'replicas only'): the task-vector
    merge, RegMean at base size (fp64 MFMA GEMMs + Cholesky solve) and the Gram capture SYRK, the last two against the fp64
    MFMA roofline."""
    import statistics
    ops = importlib.import_module("vl_merging_amd.ops")
    M = importlib.import_module("vl_merging_amd.merge")
    bm = importlib.import_module("vl_merging_amd.bench_merge")
    rg = importlib.import_module("vl_merging_amd.regmean")
    out = {}

    def ev_median(fn, reps=5, warmup=2):
        for _ in range(warmup):
            fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        return statistics.median(ts)

    try:  # task-vector merge (vilt_module.py:640-746): all_moe experts + a central ufo checkpoint -> ufo
        sd = bm.synthetic_all_moe_blocks()
        gen = torch.Generator(device="cuda").manual_seed(7)
        central = {}
        for k, v in sd.items():
            dst = k.replace(".v.", ".").replace(".l.", ".").replace(".vl.", ".")
            if dst not in central:
                central[dst] = torch.randn(v.shape, device="cuda", generator=gen) * 0.02
        cfg = dict(vlffn_start_layer_index=10, only_activate_used_experts=False, sum_lambda=0.75, loss_names={})
        plans = []
        M.sum_task_vectors(sd, cfg, central_weight=central, plan_out=plans)
        plan = plans[0]
        t = ev_median(plan.run, reps=10, warmup=3)
        nbytes = plan.bytes_read + plan.bytes_written
        out["task_vector"] = {"GBps": nbytes / t / 1e9, "seconds_median": t, "algorithmic_bytes": nbytes,
                              "frac_of_hbm_peak": nbytes / t / 1e9 / HBM_PEAK_GBPS}
        del central, plans, plan
    except Exception as e:
        out["task_vector"] = {"error": repr(e)}
    try:  # RegMean at base size, device-resident inputs (vilt_module.py:366-531): W.G GEMMs, Cholesky, two triangular solves
        gen = torch.Generator(device="cuda").manual_seed(3)
        grams = {}
        for k, v in sd.items():
            if k.endswith(".weight") and "norm" not in k and ".vl." not in k:
                name = k.replace(".qkv.weight", "") if "qkv" in k else k.replace(".weight", "")
                D = v.shape[1]
                x = torch.randn(D + 64, D, device="cuda", dtype=torch.float64, generator=gen)
                grams[name] = x.t() @ x
        cfg = dict(vlffn_start_layer_index=10, only_activate_used_experts=False, scaling_for_non_diag=0.9,
                   loss_names={"irtr": 1}, gram_matrices=None)
        res = rg.regmean(sd, cfg, gram_matrices=grams)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            res = rg.regmean(sd, cfg, gram_matrices=grams)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = statistics.median(ts)
        # per merged weight [out, in] of two experts: 2 products W_m G'_m (2 out in^2 each), Cholesky in^3 / 3, two triangular
        # solves of `out` right-hand sides (out in^2 each)
        fl = sum(2 * 2.0 * v.shape[0] * v.shape[1] ** 2 + v.shape[1] ** 3 / 3.0 + 2.0 * v.shape[0] * v.shape[1] ** 2
                 for k, v in res.items() if k.endswith(".weight") and v.dim() == 2 and "norm" not in k and "blocks" in k)
        out["regmean_base"] = {"seconds": dt, "seconds_min_max": [min(ts), max(ts)], "fp64_flop": fl, "fp64_tflops": fl / dt / 1e12,
                               "frac_of_fp64_mfma_peak": fl / dt / 1e12 / F64_MFMA_PEAK_TFLOPS, "peak_tflops": F64_MFMA_PEAK_TFLOPS,
                               "note": "wall time of the whole merge (48 weight solves + HIP averages, ~1000 launches from Python); "
                                       "the GEMM kernel alone: gram_capture_base"}
        del grams, res, sd
    except Exception as e:
        out["regmean_base"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    try:  # Gram capture (cache_gram_matrices.py:246-254): G += X^T X in fp64 on the device, X = one hooked input of the 88-sample pass
        legs = {}
        for D in (768, 3072):
            rows = 54296
            x = torch.randn(rows, D, device=dev).to(torch.bfloat16)
            g = torch.zeros(D, D, device=dev, dtype=torch.float64)
            t = ev_median(lambda: ops.gram_accumulate(x, g), reps=5, warmup=1)
            fl = 2.0 * rows * D * D  # the reference's X^T X (torch.matmul computes the full product; the SYRK does half of it)
            legs["D%d" % D] = {"seconds_median": t, "rows": rows, "tflops_over_2MD2": fl / t / 1e12,
                               "frac_of_fp64_mfma_peak": fl / t / 1e12 / F64_MFMA_PEAK_TFLOPS}
            del x, g
        out["gram_capture_base"] = dict(legs, peak_tflops=F64_MFMA_PEAK_TFLOPS, kernel="gram_f64_kernel")
    except Exception as e:
        out["gram_capture_base"] = {"error": repr(e)}
    torch.cuda.empty_cache()
    return out


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher (reference: `run.py:263-288`, `gpus=N, accelerator="ddp"` makes
    Lightning start one process per GPU): this parent starts N fresh rank processes BEFORE anything here touches the
    GPU, stays off the GPU itself, relays rank 0's JSON line as its own last line and fails if any rank fails.
    Every rank's stderr is relayed line by line with a `[rank r]` tag; a wall-clock limit (VLM_BENCH_TIMEOUT_S, default
    1800 s) ends a job whose ranks are all alive but stuck in a collective."""
    import subprocess
    import threading
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    limit = float(os.environ.get("VLM_BENCH_TIMEOUT_S", "1800"))
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE, text=True))
    buf = []

    def relay(r, stream):
        for ln in stream:
            sys.stderr.write("[rank %d] %s" % (r, ln if ln.endswith("\n") else ln + "\n"))
            sys.stderr.flush()

    threads = [threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)]
    threads += [threading.Thread(target=relay, args=(r, p.stderr), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    codes = [None] * n
    t_start = time.time()
    timed_out = False
    while any(c is None for c in codes):  # a rank that dies leaves its peers waiting in a collective: end them (our own PIDs)
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        timed_out = time.time() - t_start > limit
        if any(c for c in codes if c is not None) or timed_out:
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.kill()
                    codes[i] = p.wait()
            break
        time.sleep(0.2)
    for t in threads:
        t.join(timeout=30)
    out0 = buf[0] if buf else ""
    lines = [ln for ln in out0.splitlines() if ln.strip()]
    for ln in lines[:-1]:
        sys.stderr.write("[rank 0 stdout] " + ln + "\n")
    if timed_out:
        sys.stderr.write("bench.py: no result after %.0f s (VLM_BENCH_TIMEOUT_S): ranks killed, exit codes %r\n" % (limit, codes))
        sys.exit(124)
    if any(codes):
        sys.stderr.write("bench.py: rank exit codes %r\n" % (codes,))
        if lines:
            sys.stderr.write(lines[-1] + "\n")
        sys.exit(next(c for c in codes if c) or 1)
    line = lines[-1] if lines else ""
    got = json.loads(line)  # rank 0 must have produced the JSON line
    assert got["n_gpus"] == n, "rank 0 reported n_gpus=%r for --gpus %d" % (got.get("n_gpus"), n)
    sys.stdout.write(line + "\n")
    sys.stdout.flush()


def dry_run(args, rank, world):
    """VLM_BENCH_DRY_RUN=1: rendezvous + one MAX all-reduce over gloo on the CPU, no model, no GPU (launcher test)."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert float(t) == world
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        _, _, _, cfg_index, task_text = TASKS[args.task]
        print(json.dumps({"metric": "dry run (launcher + rendezvous only)", "value": None, "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "dry_run": True, "task": task_text, "arch": args.arch,
                          "per_gpu_batchsize": args.batch, "baseline_config": cfg_index[args.arch]}))


def parity_gates(dev, mods, cpu, merge_check):
    """The checks that stand beside the numbers in the JSON line: (a) the merged buffer bench_merge
    has just timed against oracle/merge_oracle.py on a slice of its tensors, bit for bit; (b) the loss of one configs[0] step
    (base_vl ufo 224^2 B = 2, eval mode: B = 2 forces the hard negatives) on the GPU against the oracle's loss on the same
    weights and batch -- the oracle step cpu_baseline() runs anyway.  The oracle is the CHECKER here, never the thing timed."""
    out = {}
    if merge_check is not None:
        try:
            import numpy as np
            from oracle import merge_oracle as mo
            sub, got, cfg, layers = merge_check
            ref = mo.merge_weights({k: v.cpu().numpy() for k, v in sub.items()}, cfg, layers=layers)
            bad = [k for k, v in got.items() if v.cpu().numpy().tobytes() != np.asarray(ref[k]).tobytes()]
            out["merge"] = "bit-exact" if not bad else "MISMATCH: " + ", ".join(bad[:3])
            out["merge_checked"] = "%d merged tensors of layers 0 and 11 (%d bytes) against oracle/merge_oracle.py" % (
                len(got), sum(v.numel() * 4 for v in got.values()))
        except Exception as e:
            out["merge"] = "error: " + repr(e)
    if cpu is not None and cpu.get("_sd") is not None:
        try:
            cfgmod, vm = mods["cfg"], mods["vm"]
            cfg = cfgmod.make_config("task_mlm_itm_ifm_square_randaug_base_vl", "ufo", per_gpu_batchsize=2, num_gpus=1)
            torch.manual_seed(0)
            model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
            info = model.load_state_dict({k: v.detach() for k, v in cpu["_sd"].items()}, strict=False)
            model = model.to(dev).eval()
            model.setup_engine()
            b = cpu["_batch"]
            batch = {k: ([v[0].to(dev)] if k == "image" else v.to(dev)) for k, v in b.items()}
            with torch.no_grad():
                got = float(model.training_step({"vl": batch}, 0))
            want = cpu["_oracle_loss"]
            out.update(step_loss_gpu=got, step_loss_oracle=want, step_loss_err=abs(got - want), step_loss_tol=3e-2,
                       step_loss_ok=bool(abs(got - want) <= 3e-2),
                       step="BASELINE configs[0]: base_vl ufo 224^2 T=40 B=2 mlm+itm+ifm, eval mode, same weights and batch; "
                            "bf16 MFMA engine vs fp32 oracle (tolerance as tests/test_model_gpu.py: losses 3e-2)",
                       step_unloaded_keys=len(info.missing_keys))
            del model
            torch.cuda.empty_cache()
        except Exception as e:
            out["step_loss_err"] = None
            out["step_error"] = repr(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--task", default="pretrain", choices=sorted(TASKS),
                    help="pretrain = task_mlm_itm_ifm (BASELINE configs[1] ufo / [2] all_moe); irtr = task_finetune_irtr_coco "
                         "(configs[4]: `--gpus 8 --task irtr` is that config as BASELINE.json words it, per-GPU batch 20)")
    ap.add_argument("--arch", default="ufo", choices=["ufo", "all_moe"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 22 pretrain, 20 irtr)")
    ap.add_argument("--image-size", type=int, default=384)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-merge", action="store_true")
    ap.add_argument("--no-gemm-timer", action="store_true", help="skip the per-launch HIP events (overhead check)")
    ap.add_argument("--gemm-timer-every", type=int, default=8,
                    help="bracket the GEMM launches of every n-th timed step with HIP events.  A bracketed step also runs "
                         "without the weight-gradient side stream (a launch's duration must be its own) and costs 4 ms more "
                         "than the 65.5 ms of an ordinary step (step_ms_rank0 shows it): every 8th keeps that at 0.8 %% of "
                         "`value`; one step is 340 bracketed launches")
    ap.add_argument("--no-calibrate", action="store_true", help="skip the attainable-peak probes")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs' legs (the other two DP "
                                                                "workloads at this N; at N = 1 also task vector, RegMean, Gram capture)")
    args = ap.parse_args()
    # host-side torch ops (model construction, the oracle) on every logical CPU of a 256-thread host are slower than on 16
    torch.set_num_threads(min(torch.get_num_threads(), 16))
    if args.batch is None:
        args.batch = TASKS[args.task][1]

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (the number is whole-job: they must agree)" % (args.gpus, world))
    if os.environ.get("VLM_BENCH_DRY_RUN", "0") != "0":
        if os.environ.get("VLM_BENCH_TEST_FAIL_RANK") == str(rank):  # launcher test: a rank that dies before rendezvous
            sys.exit(3)
        if os.environ.get("VLM_BENCH_TEST_HANG"):  # launcher test: every rank alive, none making progress
            sys.stderr.write("hanging on purpose\n")
            sys.stderr.flush()
            time.sleep(3600)
        return dry_run(args, rank, world)
    if rank != 0:  # RCCL prints a version banner on stdout in every process: only rank 0 may write there
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    # VLM_BENCH_ONE_DEVICE=1 + VLM_DIST_BACKEND=gloo: several ranks on ONE GPU (a functional test of the multi-rank
    # path on a single-GPU box; RCCL itself refuses two ranks per device, gloo stages CUDA tensors through the host)
    if os.environ.get("VLM_BENCH_ONE_DEVICE", "0") != "0":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("VLM_BENCH_FORCE_DIST", "0") != "0"  # 1-GPU smoke test of the RCCL code path
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        backend = os.environ.get("VLM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    ge.import_package()
    mods = {"cfg": importlib.import_module("vl_merging_amd.vilt.config"),
            "vm": importlib.import_module("vl_merging_amd.vilt.modules.vilt_module"),
            "vu": importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils"),
            "ddp": importlib.import_module("vl_merging_amd.ddp")}
    ops = importlib.import_module("vl_merging_amd.ops")
    L_ = importlib.import_module("vl_merging_amd._lib")
    eng = importlib.import_module("vl_merging_amd.engine")

    if rank == 0:
        _phase("building %s %s B=%d %d^2 at world %d" % (args.task, args.arch, args.batch, args.image_size, world))
    leg = TrainLeg(mods, args.task, args.arch, args.batch, args.image_size, rank, world, dev, force_dist)
    reducer = leg.reducer
    timer = GemmTimer(ops)
    timer.install()
    atimer = AttnTimer(ops, timer)
    atimer.install()
    use_timer = rank == 0 and not args.no_gemm_timer
    # The steps whose GEMM launches are bracketed run with the weight-gradient side stream OFF: a launch that shares the chip
    # with another stream's kernel takes longer for reasons that are not its own, and the roofline figure is the kernel's own
    # duration.  (They stay inside the timed region: `value` pays for them.)
    side_default = eng._WGRAD["enabled"]

    def per_step(it):
        timer.on = use_timer and it % max(1, args.gemm_timer_every) == 0
        eng._WGRAD["enabled"] = side_default and not timer.on

    if rank == 0:
        _phase("timed region: %d warm-up + %d steps" % (args.warmup, args.steps))
    dt, loss = leg.timed(args.steps, args.warmup, dev, per_step)
    if rank == 0:
        _phase("timed region done: %.2f ms per step" % (dt / args.steps * 1e3))
    eng._WGRAD["enabled"] = side_default
    timer.on = False
    exposed_comm_ms = reducer.exposed_wait_ms() / max(1, args.steps)
    loss_val = float(loss.detach())
    finite = math.isfinite(loss_val)
    if world > 1:  # every rank takes the same decision (a rank that raised alone would leave the others in their next collective)
        flag = torch.tensor([1.0 if finite else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        finite = bool(flag.item() > 0.5)
    if not finite:
        # a step whose activations are NaN runs FASTER (less switching power, higher clocks: docs/experiments.md round 6): a rate
        # measured on such a run is not a measurement
        raise RuntimeError("bench: the last timed step's loss is %r -- the timed steps did not compute the workload" % loss_val)
    total_samples = args.batch * world * args.steps
    value = total_samples / dt
    _, _, flop384, cfg_index, task_text = TASKS[args.task]

    out = None
    if rank == 0:
        gs = timer.summary()
        out = {
            "metric": "VL samples/sec (fwd+bwd) base_vl 384^2 at 1/2/4/8 GPUs; merge GB/s vs HBM peak",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "step_ms_rank0": [round(x, 2) for x in leg.step_ms],
            "ms_per_step_median": sorted(leg.step_ms)[len(leg.step_ms) // 2], "setup_steps": SETUP_STEPS,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "base_vl %s %d^2 patch16 T=40 per_gpu_batchsize=%d %s fwd+bwd+AdamW, "
                                   "train mode (BASELINE configs[%d])" % (args.arch, args.image_size, args.batch, task_text,
                                                                          cfg_index[args.arch]),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "final_loss": loss_val},
            # time the compute stream waits for the tail of the gradient all-reduce (rank 0), per step
            "exposed_comm_ms_per_step": exposed_comm_ms,
            "grad_comm": {"dtype": "bf16" if reducer.comm_dtype is not None else "fp32", "collective": reducer.collective,
                          "sharded_optimizer": reducer.sharded, "bytes_per_step": int(reducer.flat.numel) * (2 if reducer.comm_dtype is not None else 4),
                          "buckets": reducer.bucket_plan(),
                          "cu_budget": L_.get_lib().vlm_device_cus(),  # VLM_GEMM_CUS: CUs the GEMM grids plan for (RCCL takes the rest)
                          "cu_budget_set_by_reducer": reducer.cu_budget_set,
                          # self-diagnosis of the first multi-GPU run: which library moves the gradients, between how many
                          # ranks, and when each bucket of the last timed step was issued / complete (ms from the step's start)
                          "backend": (dist.get_backend() if dist.is_initialized() else None),
                          "rccl_ranks": (dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0),
                          "bucket_timeline_last_step": reducer.bucket_timeline(),
                          "wgrad_side_stream": bool(side_default), "standin": reducer.standin},
        }
        if args.image_size == 384:
            flop_per_sample = flop384
        else:
            flop_per_sample = 657.5e9 if args.task == "pretrain" else 3 * 41.98e9  # 224^2
        out["model_tflops"] = value * flop_per_sample / 1e12 / world
        if gs:
            n_timed = len(range(0, args.steps, max(1, args.gemm_timer_every)))
            traffic, traffic_meta = pmc_traffic(GEMM_KERNELS, GEMM_HELPERS, with_meta=True)
            out["roofline"] = {"bound": "mfma", "achieved": gs["tflops"], "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": gs["tflops"] / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_meta,
                               "traffic_unit": "HBM-side bytes per vlm_gemm_bf16 call, launch-weighted over the kernels "
                                               "that serve it (rocprofv3 PMC, the newest profiles/r*_pmc_traffic.json)",
                               "kernel": "vlm_gemm_bf16 (+ _grouped / vlm_gemm_wgrad_grouped for all_moe): " + " / ".join(GEMM_KERNELS),
                               "note": "peak = nominal dense bf16; a loop of nothing but independent MFMAs reaches 1515 "
                                       "TFLOP/s on this chip (clock drops to 1.45 GHz: DESIGN.md 4.1); the bracketed steps run "
                                       "with the weight-gradient side stream off (every launch alone on the chip)",
                               "launches": gs["launches"], "avg_launch_us": gs["avg_us"], "timed_steps": n_timed,
                               "gemm_share_of_step": gs["seconds"] / (dt / args.steps * n_timed)}
            ats = atimer.summary()
            if ats:
                # the kernels north_star puts a number on, timed in this run (not only in profiles/): same bracketed steps
                out["roofline_attention"] = ats
        if not args.no_calibrate:
            out["attainable"] = calibrate(dev)
    leg.close()
    del leg, reducer

    secondary = {}
    small = os.environ.get("VLM_BENCH_FORCE_SECONDARY", "0") != "0"  # tests: the legs at the headline's (small) geometry
    if not args.no_secondary and (args.image_size == 384 or small):
        # every rank takes part: the other two data-parallel BASELINE workloads at this world size
        if rank == 0:
            _phase("secondary data-parallel legs")
        secondary.update(secondary_train_legs(mods, dev, rank, world, (args.task, args.arch), force_dist,
                                              image_size=args.image_size, batch=args.batch if small else None))
    if rank == 0 and secondary and out is not None:
        out["secondary"] = dict(secondary)
    if rank == 0:
        merge_check = None
        if not args.no_merge:
            _phase("merge leg")
            bm = importlib.import_module("vl_merging_amd.bench_merge")
            m = bm.run(check_layers=(0, 11))
            merge_check = m.pop("check")
            out["merge"] = {"metric": "all_moe->ufo interpolation merge, base size, fp32", "GBps": m["GBps"],
                            "seconds_median": m["seconds_median"], "algorithmic_bytes": m["algorithmic_bytes"],
                            "roofline": {"bound": "hbm", "achieved": m["GBps"], "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                         "frac": m["GBps"] / HBM_PEAK_GBPS,
                                         "traffic": pmc_traffic("vlm_merge_kernel"), "kernel": "vlm_merge_kernel"}}
        if not args.no_secondary and world == 1 and not force_dist and args.image_size == 384:
            _phase("secondary merge legs (task vector, RegMean, Gram capture)")
            secondary.update(secondary_benches(dev))
        if secondary:
            out["secondary"] = secondary
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            _phase("CPU baseline (oracle on the host cores)")
            try:
                cpu = cpu_baseline()
            except Exception as e:  # the baseline must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        if merge_check is not None or cpu is not None:
            _phase("parity gates")
            out["parity"] = parity_gates(dev, mods, cpu, merge_check)
        _phase("done")
        if cpu is not None:
            out["cpu_baseline"] = {k: v for k, v in cpu.items() if not k.startswith("_")}
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes last: whatever native libraries (RCCL's banner) left in C stdio buffers is flushed first
        sys.stdout.flush()
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.write(json.dumps(out) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
