"""Checkpoint merging on the GPU: interpolation, task-vector arithmetic, RegMean.

Drop-in for ViLTransformerSS.merge_weights / sum_task_vectors / regmean
(reference src/vilt/modules/vilt_module.py:533-638, :640-746, :366-531): same `state_dict -> state_dict`
contract, same key grammar, same pass-through-by-identity of non-block keys, same "already merged key
passes through" and KeyError behaviour; the per-element arithmetic of all 156 output tensors runs in ONE
launch of the HIP merge kernel (csrc/merge.hip) and is bit-exact with the reference's CPU result.
"""
import ctypes
from typing import Dict, List, Optional

import torch

from . import _lib as L

# (source template, destination template): vilt_module.py:543-551
_LAYERS = [
    ("transformer.blocks.{i}.attn.{m}.qkv.weight", "transformer.blocks.{i}.attn.qkv.weight", (None,)),
    ("transformer.blocks.{i}.attn.{m}.proj.{n}", "transformer.blocks.{i}.attn.proj.{n}", ("weight", "bias")),
    ("transformer.blocks.{i}.attn.{m}.{n}", "transformer.blocks.{i}.attn.{n}", ("q_bias", "v_bias")),
    ("transformer.blocks.{i}.mlp.{m}.fc1.{n}", "transformer.blocks.{i}.mlp.fc1.{n}", ("weight", "bias")),
    ("transformer.blocks.{i}.mlp.{m}.fc2.{n}", "transformer.blocks.{i}.mlp.fc2.{n}", ("weight", "bias")),
    ("transformer.blocks.{i}.norm1.{m}.{n}", "transformer.blocks.{i}.norm1.{n}", ("weight", "bias")),
    ("transformer.blocks.{i}.norm2.{m}.{n}", "transformer.blocks.{i}.norm2.{n}", ("weight", "bias")),
]
NUM_MERGE_LAYERS = 12  # the reference hard-codes range(12) (:395, :553, :665)


def _tensor_names(i):
    for src_t, dst_t, leaves in _LAYERS:
        for n in leaves:
            yield (lambda m, s=src_t, n=n: s.format(i=i, m=m, n=n)), dst_t.format(i=i, n=n)


def modalities_for_layer(config, i, honour_only_used=True):
    """Which experts feed layer i (vilt_module.py:557-567; regmean's variant :397-404)."""
    loss = config["loss_names"]
    if i < config["vlffn_start_layer_index"]:
        return ["v", "l"]
    if honour_only_used:
        if config["only_activate_used_experts"]:
            if loss.get("irtr", 0) > 0:
                return ["v", "l"]
            if loss.get("vqa", 0) > 0 or loss.get("nlvr2", 0) > 0:
                return ["vl"]
            # the reference leaves modalities=None and dies at len(None) (:569); same error class
            raise TypeError("object of type 'NoneType' has no len()")
        return ["v", "l", "vl"]
    if loss.get("irtr", 0) > 0:
        return ["v", "l"]
    if loss.get("vqa", 0) > 0:
        return ["vl"]
    return ["v", "l", "vl"]


def interpolation_ratios(modalities, merge_ratio):
    """vilt_module.py:569-584 (python doubles; rounded to fp32 when they meet the tensor)."""
    if len(modalities) == 1:
        return {modalities[0]: 1}
    if len(modalities) == 3:
        return {"v": (2 / 3) * merge_ratio, "l": (2 / 3) * (1 - merge_ratio), "vl": 1 / 3}
    return {"v": merge_ratio, "l": 1 - merge_ratio}


class MergePlan:
    """A device-resident job table for csrc/merge.hip; build once, run() launches one kernel."""

    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.VlmError("the merge kernel runs on the GPU only (got device %s)" % device)
        self.jobs: List[L.MergeJob] = []
        self.keep = []  # keeps staged tensors alive
        self.total = 0
        self.bytes_read = 0
        self.bytes_written = 0
        self.ws = None

    def _dev(self, t):
        t = t.detach()
        if t.dtype != torch.float32:
            raise L.VlmError("merge expects float32 checkpoints, got %s" % t.dtype)
        if t.device != self.device or not t.is_contiguous() or (t.data_ptr() & 15):
            t = t.to(self.device, copy=True).contiguous()
        self.keep.append(t)
        return t

    def add(self, mode, srcs, ratios, base=None, out=None):
        srcs = [self._dev(s) for s in srcs]
        n = srcs[0].numel()
        for s in srcs:
            if s.shape != srcs[0].shape:
                raise L.VlmError("merge sources disagree in shape: %s vs %s" % (s.shape, srcs[0].shape))
        if out is None:
            out = torch.empty_like(srcs[0])
        self.keep.append(out)
        job = L.MergeJob()
        job.dst = out.data_ptr()
        job.base = 0
        if mode == L.MERGE_TASKVEC:
            b = self._dev(base)
            job.base = b.data_ptr()
            self.bytes_read += 4 * n
        for k, s in enumerate(srcs):
            job.src[k] = s.data_ptr()
            job.ratio[k] = float(ratios[k]) if ratios is not None else 1.0
        job.n_src = len(srcs)
        job.mode = mode
        job.n_elem = n
        self.jobs.append(job)
        self.total += n
        self.bytes_read += 4 * n * len(srcs)
        self.bytes_written += 4 * n
        return out

    def upload(self):
        lib = L.get_lib()
        n = len(self.jobs)
        arr = (L.MergeJob * n)(*self.jobs)
        nbytes = lib.vlm_merge_plan_bytes(n, self.total)
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            L.check(lib.vlm_merge_plan_upload(arr, n, L.ptr(self.ws), nbytes, L.stream_ptr()), "vlm_merge_plan_upload")
        return self

    def run(self):
        if self.ws is None:
            self.upload()
        with torch.cuda.device(self.device):
            L.check(L.get_lib().vlm_merge_run(L.ptr(self.ws), L.stream_ptr()), "vlm_merge_run")


def _passthrough(state_dict):
    # vilt_module.py:537-541: same tensor objects, not copies
    return {k: v for k, v in state_dict.items() if "transformer.blocks." not in k or "gamma" in k}


def _collect(state_dict, src, dst, modalities):
    """Return (list of present sources | None, passthrough tensor | None) with the reference's break rule."""
    srcs = []
    for m in modalities:
        name = src(m)
        if name in state_dict:
            srcs.append((m, state_dict[name]))
        else:
            return None, state_dict[dst]  # KeyError if neither exists, as in the reference (:597-599)
    return srcs, None


def merge_weights(state_dict: Dict[str, torch.Tensor], config, device="cuda", plan_out: Optional[list] = None):
    """Interpolation merge (vilt_module.py:533-638)."""
    out = _passthrough(state_dict)
    plan = MergePlan(device)
    for i in range(NUM_MERGE_LAYERS):
        mods = modalities_for_layer(config, i)
        ratios = interpolation_ratios(mods, config["merge_ratio"])
        for src, dst in _tensor_names(i):
            srcs, through = _collect(state_dict, src, dst, mods)
            if srcs is None:
                out[dst] = through
            else:
                out[dst] = plan.add(L.MERGE_LERP, [t for _, t in srcs], [ratios[m] for m, _ in srcs])
    if plan.jobs:
        plan.run()
    if plan_out is not None:
        plan_out.append(plan)
    return out


def sum_task_vectors(state_dict, config, central_weight=None, device="cuda", plan_out: Optional[list] = None):
    """Task-vector merge (vilt_module.py:640-746).  `central_weight` defaults to torch.load(config[...])."""
    out = _passthrough(state_dict)
    if central_weight is None:
        from . import checkpoint
        central_weight = checkpoint.load_file(config["central_weight"])
    if "state_dict" in central_weight:
        central_weight = central_weight["state_dict"]
    plan = MergePlan(device)
    lam = config["sum_lambda"]
    for i in range(NUM_MERGE_LAYERS):
        mods = modalities_for_layer(config, i)
        for src, dst in _tensor_names(i):
            central = central_weight[dst]
            srcs, through = _collect(state_dict, src, dst, mods)
            if srcs is None:
                out[dst] = through
            else:
                r = [1 if len(mods) == 1 else lam] * len(srcs)
                out[dst] = plan.add(L.MERGE_TASKVEC, [t for _, t in srcs], r, base=central)
    if plan.jobs:
        plan.run()
    if plan_out is not None:
        plan_out.append(plan)
    return out
