"""Optimizer / schedule of the hot path (reference src/vilt/modules/vilt_utils.py:218-359): set_task, the four
parameter groups of set_schedule, HF-4.x AdamW and the polynomial-decay-with-warmup schedule, executed as ONE fused
HIP AdamW launch per parameter group over the engine's flat buffers.
"""
import re

import torch

from ... import engine
from ... import ops

NO_DECAY = ["bias", "LayerNorm.bias", "LayerNorm.weight", "norm.bias", "norm.weight", "norm1.bias", "norm1.weight",
            "norm2.bias", "norm2.weight", "norm.v.bias", "norm.v.weight", "norm.l.bias", "norm.l.weight",
            "norm.vl.bias", "norm.vl.weight"]  # vilt_utils.py:230-246


def set_task(pl_module):
    pl_module.current_tasks = [k for k, v in pl_module.hparams.config["loss_names"].items() if v >= 1]


def head_names(config):
    names = ["vqa_classifier", "nlvr2_classifier", "img_cls_classifier"]
    if config["all_mlp_mult"]:
        names.append("mlp")
    if config["all_vl_mult"]:
        names += ["attn.vl", "mlp.vl", "mlp_vl"]
    if config["all_v_mult"]:
        names += ["attn.v", "mlp.v"]
    if config["all_l_mult"]:
        names += ["attn.l", "mlp.l"]
    return names


def param_group_of(name, heads):
    """0: decay/body, 1: no-decay/body, 2: decay/head, 3: no-decay/head (vilt_utils.py:272-312)."""
    nd = any(s in name for s in NO_DECAY)
    hd = any(s in name for s in heads)
    return (2 if hd else 0) + (1 if nd else 0)


_BLOCK = re.compile(r"transformer\.blocks\.(\d+)\.")


def flat_order_key(name):
    """Flat-buffer order: (optimizer group is resolved later) embeddings/table first, then blocks 0..L-1, then heads:
    the reverse of the order in which backward finishes them, so DDP buckets are contiguous slices."""
    m = _BLOCK.search(name)
    if m:
        # inside a block: decay group first, then no-decay -> two AdamW ranges per block instead of one per tensor
        return (1, int(m.group(1)), param_group_of(name, ["vqa_classifier", "nlvr2_classifier", "img_cls_classifier"]),
                name)
    early = ("text_embeddings", "token_type_embeddings", "transformer.patch_embed", "transformer.cls_token",
             "transformer.mask_token", "relative_position_bias_table", "temporal_relative_position_bias_table")
    g = param_group_of(name, ["vqa_classifier", "nlvr2_classifier", "img_cls_classifier"])
    if any(name.startswith(e) for e in early):
        return (0, 0, g, name)
    return (2, 0, g, name)


def polynomial_decay_lambda(step, warmup, total, lr_init, lr_end=0.0, power=1.0):
    """transformers.get_polynomial_decay_schedule_with_warmup's lr_lambda."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    if step > total:
        return lr_end / lr_init
    rem = 1 - (step - warmup) / (total - warmup)
    return ((lr_init - lr_end) * rem ** power + lr_end) / lr_init


class FusedAdamW:
    """HF-4.x AdamW semantics (vilt_utils.py:314-317) over the flat buffers; param_groups mirrors torch optimizers."""

    def __init__(self, pl_module, lr, betas, eps, groups):
        self.flat = pl_module._flat
        self.model = pl_module
        self.betas, self.eps = betas, eps
        self.param_groups = groups  # dicts: lr, weight_decay, initial_lr, ranges [(start, end)]
        self.step_count = 0
        self.m = torch.zeros_like(self.flat.flat_p)
        self.v = torch.zeros_like(self.flat.flat_p)
        self._shard = None        # sharded mode: [(lo, hi, state offset)] of the flat ranges this rank owns
        self._after_step = None   # sharded mode: all-gather of the updated parameters
        self.grad_scale = 1.0
        self.tail_sync = None
        # HF AdamW skips parameters whose .grad is None (`if p.grad is None: continue`): tensors no pass of the task set
        # touches (deep v / l experts under VQA, vl experts under irtr, mask_token, position_embeddings ...) get neither
        # an Adam update nor weight decay and stay bit-constant.  "Has a gradient" is STRUCTURAL here, as in torch: the
        # set of parameters some backward pass has written so far (engine.FlatParams.touched), never a test of gradient
        # values -- a tensor whose gradient is exactly zero (DropPath dropped its branch for the whole batch) is still
        # updated (weight decay) like the reference's zero-but-not-None .grad.  Once reached a tensor stays active
        # (torch keeps a zeroed .grad after zero_grad()).  Rank-independent: every rank runs the same passes.
        self._group_of = {}
        for gi, g in enumerate(groups):
            for lo, hi in g["ranges"]:
                for n in self.flat.names:
                    o, _ = self.flat.offsets[n]
                    if lo <= o < hi:
                        self._group_of[n] = gi
        self._all_ranges = [list(g["ranges"]) for g in groups]
        self._active = set()

    def set_shard(self, ranges, after_step):
        """ddp_sharded (reference run.py:231-232): this rank updates only `ranges` (its chunk of every gradient bucket,
        ddp.FlatGradReducer.own_ranges) and keeps Adam's m / v for those elements only, packed back to back;
        `after_step` all-gathers the updated fp32 parameters."""
        if self.step_count:
            raise RuntimeError("set_shard() must come before the first optimizer step")
        off, table = 0, []
        for lo, hi in sorted(ranges):
            table.append((lo, hi, off))
            off += hi - lo
        self._shard = table
        self._after_step = after_step
        dev = self.flat.flat_p.device
        self.m = torch.zeros(off, device=dev, dtype=self.flat.flat_p.dtype)
        self.v = torch.zeros(off, device=dev, dtype=self.flat.flat_p.dtype)

    def state_elements(self):
        return self.m.numel()

    def _discover_active(self):
        """Re-derive each group's ranges from the parameters a backward pass has written so far (no device work)."""
        f = self.flat
        self._active |= f.touched
        per_group = [[] for _ in self.param_groups]
        for n in f.names:
            if n in self._active and n in self._group_of:
                o, _ = f.offsets[n]
                per_group[self._group_of[n]].append((o, o + f.extent[n]))
        for g, r in zip(self.param_groups, per_group):
            g["ranges"] = _merge_ranges(r)

    def inactive_parameters(self):
        return [n for n in self.flat.names if n not in self._active]

    # ---- checkpoint / resume (Lightning keeps `optimizer_states` in the .ckpt, run.py:189-195, :218-223, :280) --------
    def _torch_order(self):
        """(index -> parameter name, per-group index lists) in the order torch numbers the reference's optimizer: the four
        groups of set_schedule one after the other, each in named_parameters() order (vilt_utils.py:272-312)."""
        heads = head_names(self.model.hparams.config)
        per_group = [[] for _ in range(4)]
        for n, _ in self.model.named_parameters():
            per_group[param_group_of(n, heads)].append(n)
        names = [n for g in per_group for n in g]
        idx, k = [], 0
        for g in per_group:
            idx.append(list(range(k, k + len(g))))
            k += len(g)
        return names, idx

    def _full_moments(self):
        """Adam's m / v over the whole flat buffer.  Sharded mode (a COLLECTIVE: every rank calls it): each rank scatters the
        chunks it owns into zeros and the ranks' disjoint pieces meet by all-reduce."""
        if self._shard is None:
            return self.m, self.v
        import torch.distributed as dist
        f = self.flat
        full = []
        for packed in (self.m, self.v):
            t = torch.zeros_like(f.flat_p)
            for lo, hi, off in self._shard:
                t[lo:hi].copy_(packed[off:off + hi - lo])
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(t)
            full.append(t)
        return full

    def state_dict(self):
        """torch.optim layout, as the reference's `resume_from_checkpoint` reads it (its optimizer is transformers.AdamW, a
        torch.optim.Optimizer: state keyed by the parameter's index in param_groups order with `step` / `exp_avg` /
        `exp_avg_sq`, param_groups with `params` index lists).  Only parameters that have an Adam state appear (HF AdamW
        creates it at a parameter's first gradient).  `vlm_*` keys carry what this engine needs on top: the index -> name map
        (a resume does not depend on registration order) and the set of parameters a backward pass has reached.
        With a sharded optimizer this is a collective (every rank calls it; every rank gets the full state)."""
        f = self.flat
        m, v = self._full_moments()
        names, idx = self._torch_order()
        state = {}
        for i, n in enumerate(names):
            if n not in self._active or n not in f.offsets:
                continue
            o, k = f.offsets[n]
            state[i] = {"step": int(self.step_count), "exp_avg": m[o:o + k].detach().cpu().clone(),
                        "exp_avg_sq": v[o:o + k].detach().cpu().clone()}
        shapes = {n: tuple(p.shape) for n, p in self.model.named_parameters()}
        for i, st in state.items():
            st["exp_avg"] = st["exp_avg"].view(shapes[names[i]])
            st["exp_avg_sq"] = st["exp_avg_sq"].view(shapes[names[i]])
        groups = []
        for g, ids in zip(self.param_groups, idx):
            d = {k: v_ for k, v_ in g.items() if k != "ranges"}
            d.update(betas=tuple(self.betas), eps=self.eps, correct_bias=True, params=ids)
            groups.append(d)
        return {"state": state, "param_groups": groups, "vlm_names": names, "vlm_step": int(self.step_count),
                "vlm_active": sorted(self._active)}

    def load_state_dict(self, sd):
        """Accepts (a) the torch layout above -- written here or by the reference's own run (no vlm_* keys: the index -> name map
        is rebuilt from this model's named_parameters() and checked against the file's group sizes) -- and (b) the
        name-keyed layout round-3 checkpoints hold.  Anything else raises: a resume that silently drops Adam's moments
        restarts the warm-up on trained weights."""
        f = self.flat
        state = sd.get("state")
        if not isinstance(state, dict) or "param_groups" not in sd:
            raise ValueError("optimizer state has neither the torch.optim layout nor this engine's: keys %s" % sorted(sd)[:8])
        by_name, step = {}, sd.get("vlm_step", sd.get("step"))
        if all(isinstance(k, str) for k in state):  # (b) round-3 files
            by_name = state
        else:
            names = sd.get("vlm_names")
            if names is None:
                names, idx = self._torch_order()
                sizes = [len(g.get("params", ())) for g in sd["param_groups"]]
                if sizes != [len(i) for i in idx]:
                    raise ValueError("optimizer state was written for a different model: its parameter groups hold %s "
                                     "parameters, this model's %s" % (sizes, [len(i) for i in idx]))
            for i, st in state.items():
                if not isinstance(i, int) or i >= len(names):
                    raise ValueError("optimizer state index %r outside the %d parameters of the model" % (i, len(names)))
                by_name[names[i]] = st
        full_m, full_v = (self.m, self.v) if self._shard is None else (torch.zeros_like(f.flat_p), torch.zeros_like(f.flat_p))
        steps = []
        for n, st in by_name.items():
            if n not in f.offsets:
                continue
            o, k = f.offsets[n]
            if st["exp_avg"].numel() != k:
                raise ValueError("optimizer state of %s has %d elements, the parameter %d" % (n, st["exp_avg"].numel(), k))
            full_m[o:o + k].copy_(st["exp_avg"].reshape(-1))
            full_v[o:o + k].copy_(st["exp_avg_sq"].reshape(-1))
            if "step" in st:
                steps.append(int(st["step"]))
        if self._shard is not None:
            for lo, hi, off in self._shard:
                self.m[off:off + hi - lo].copy_(full_m[lo:hi])
                self.v[off:off + hi - lo].copy_(full_v[lo:hi])
        if step is None:
            if not steps:
                raise ValueError("optimizer state carries no step count")
            step = max(steps)  # HF AdamW counts per parameter; the bias correction here is global (all active from step 1)
        self.step_count = int(step)
        f.touched |= set(sd.get("vlm_active", sd.get("active", by_name.keys())))
        self._discover_active()
        for g, src in zip(self.param_groups, sd["param_groups"]):
            for k in ("lr", "initial_lr", "weight_decay"):
                if k in src:
                    g[k] = src[k]

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self):
        self.step_count += 1
        f = self.flat
        f.finish_layerscale()  # (normally done at the end of backward / before a bucket left: nothing pending)
        engine.sync_wgrad()
        if not f.touched <= self._active:  # a pass reached parameters no earlier step had (first step, new task set)
            self._discover_active()

        def update_one(g, lo, hi, so):
            ops.adamw_step(f.flat_p[lo:hi], f.flat_g[lo:hi], self.m[so:so + hi - lo], self.v[so:so + hi - lo],
                           f.flat_b[lo:hi], g["lr"], self.betas[0], self.betas[1], self.eps, g["weight_decay"],
                           self.step_count, grad_scale=self.grad_scale, zero_grad=True)

        def update(g, lo, hi):
            if self._shard is None:
                return update_one(g, lo, hi, lo)
            for slo, shi, soff in self._shard:  # the part of [lo, hi) this rank owns
                a, b = max(lo, slo), min(hi, shi)
                if a < b:
                    update_one(g, a, b, soff + a - slo)

        # tail_sync = (lo, hi, wait): the gradient all-reduce of flat range [lo, hi) may still be in flight (DDP reducer
        # with defer_tail); everything outside it is updated first, then `wait()`, then the range itself
        tail = self.tail_sync
        late = []
        for g in self.param_groups:
            for lo, hi in g["ranges"]:
                if tail is None or hi <= tail[0] or lo >= tail[1]:
                    update(g, lo, hi)
                    continue
                if lo < tail[0]:
                    update(g, lo, tail[0])
                if hi > tail[1]:
                    update(g, tail[1], hi)
                late.append((g, max(lo, tail[0]), min(hi, tail[1])))
        if tail is not None:
            tail[2]()
        for g, lo, hi in late:
            update(g, lo, hi)
        if self._shard is not None:
            # the other ranks' chunks: their gradients are dropped, their parameters arrive by all-gather, and the bf16
            # shadows of the whole buffer are rebuilt by one cast launch
            f.flat_g.zero_()
            self._after_step()
            ops.cast_bf16(f.flat_p[:f.numel], f.flat_b[:f.numel])
        f.refresh_transposed()  # W^T shadows of the block weights (one batched launch)
        f.dirty = False  # the kernel refreshed the bf16 shadows
        f.version += 1   # derived tables cached on the parameters' contents (engine.RelPos.dense_for) are stale now


class LambdaSchedule:
    def __init__(self, optimizer, fn):
        self.opt, self.fn, self.last = optimizer, fn, 0
        for g in optimizer.param_groups:
            g["lr"] = g["initial_lr"] * fn(0)

    def state_dict(self):
        return {"last_epoch": self.last}

    def load_state_dict(self, sd):
        self.last = int(sd["last_epoch"])
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * self.fn(self.last)

    def step(self):
        self.last += 1
        for g in self.opt.param_groups:
            g["lr"] = g["initial_lr"] * self.fn(self.last)


def _merge_ranges(ranges):
    ranges = sorted(ranges)
    out = []
    for lo, hi in ranges:
        if out and out[-1][1] == lo:
            out[-1] = (out[-1][0], hi)
        else:
            out.append((lo, hi))
    return out


def group_spec(cfg):
    """(weight_decay, lr) of the four parameter groups, vilt_utils.py:272-312."""
    lr, wd = cfg["learning_rate"], cfg["weight_decay"]
    return [(wd, lr), (0.0, lr), (cfg["weight_decay_custom_modules"], lr * cfg["lr_mult"]), (0.0, lr * cfg["lr_mult"])]


def schedule_lambda(cfg, max_steps):
    """lr factor as a function of the step, vilt_utils.py:330-352 (float warm-up = fraction of max_steps)."""
    warmup = cfg["warmup_steps"]
    if isinstance(warmup, float):
        warmup = int(max_steps * warmup)
    if cfg["decay_power"] == "cosine":
        raise NotImplementedError("cosine schedule is not used by the hot-path configs")
    lr = cfg["learning_rate"]
    return lambda s: polynomial_decay_lambda(s, warmup, max_steps, lr, cfg["end_lr"], cfg["decay_power"])


def set_schedule(pl_module, max_steps=None):
    cfg = pl_module.hparams.config
    pl_module._ensure_engine()
    lr = cfg["learning_rate"]
    heads = head_names(cfg)
    spec = group_spec(cfg)
    flat = pl_module._flat
    buckets = [[] for _ in range(4)]
    for n in flat.names:
        o, k = flat.offsets[n]
        buckets[param_group_of(n, heads)].append((o, o + flat.extent[n]))
    groups = [dict(lr=l_, initial_lr=l_, weight_decay=w_, ranges=_merge_ranges(b)) for (w_, l_), b in zip(spec, buckets)]
    if cfg["optim_type"] != "adamw":
        raise NotImplementedError("only optim_type='adamw' is on the hot path")
    optimizer = FusedAdamW(pl_module, lr, (0.9, cfg["beta_2"]), 1e-8, groups)
    if max_steps is None:
        trainer = getattr(pl_module, "trainer", None)
        max_steps = getattr(trainer, "max_steps", None) if trainer is not None else None
        if max_steps is None or max_steps == -1:
            max_steps = cfg["max_steps"]
        if max_steps is None:
            raise ValueError("max_steps must be given when the config leaves it to the dataloader length")
    sched = LambdaSchedule(optimizer, schedule_lambda(cfg, max_steps))
    return [optimizer], [{"scheduler": sched, "interval": "step"}]
