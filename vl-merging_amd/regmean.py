"""RegMean merge (reference src/vilt/modules/vilt_module.py:366-531) on the GPU.

Biases / q_bias / v_bias / LayerNorm tensors: plain averages through the HIP merge kernel (VLM_MERGE_MEAN,
bit-exact with the reference: add in order, then divide by the count).  Linear weights:
W* = (sum_m W_m G'_m)(sum_m G'_m)^-1 with G' = a*G + (1-a)*diag(G), kept in float64 like the reference.
The fp64 products run on v_mfma_f64_16x16x4_f64 (csrc/f64ops.hip: vlm_gemm_f64 takes the fp32 weight directly); the
reference's torch.inverse of the sum of Gram matrices is replaced by a blocked Cholesky factorisation (the sum of
a*G + (1-a)*diag(G) of SPD Gram matrices is SPD for 0 <= a <= 1) and two triangular solves, ops.cholesky_ /
ops.solve_spd_right_.  Same W* up to fp64 rounding (tests: 1e-8 relative against the reference's outputs).
"""
import torch

from . import _lib as L
from . import merge as M
from . import ops


def scale_gram(G, alpha):
    """:388-392"""
    return alpha * G + (1 - alpha) * torch.diag_embed(torch.diag(G))


_SIDE = {}


def _side_streams(device, n):
    pool = _SIDE.setdefault(torch.device(device).index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    return pool


class _Solves:
    """The 48 independent W* = num @ inverse(den) solves of a merge (:432-434), float64, on the device.  The sum of
    a*G + (1-a)*diag(G) over SPD Gram matrices is SPD, so each is a blocked Cholesky factorisation + two triangular solves --
    a long chain of small dependent launches (48 block columns for fc2's 3072^2 Gram sum) that leaves most of the chip idle
    when run one matrix at a time.  The solves are COLLECTED and run per shape in lock step (round 5: ops.cholesky_batched_ /
    solve_spd_right_batched_: every block step is ONE launch over all matrices of a shape -- 36 of 768^2 under three right-hand-
    side shapes, 12 of 3072^2 -- ~420 launches instead of ~5 000), their positive-definiteness verdicts read once at the end.
    A Gram sum that is rank-deficient or numerically indefinite (few capture batches: fewer rows than columns for fc2,
    scaling_for_non_diag = 1) has no Cholesky factor, while the reference's LU-based torch.inverse still returns a result: that
    case falls back, loudly, to a general float64 inverse on the device (torch.linalg.inv = hipSOLVER) followed by the MFMA-f64
    product."""

    MAX_JOBS = 256

    def __init__(self, device):
        self.device = device
        # sized up front and NEVER reallocated while factorisations are in flight (their verdicts land through the raw pointer):
        # a merge has 12 layers x 4 weights = 48 solves
        self.status = torch.zeros(self.MAX_JOBS, device=device, dtype=torch.int32)
        self.jobs = []  # (what, num (solved in place), den (factorised in place), den copy, num copy)

    def submit(self, num, den, what):
        if len(self.jobs) >= self.MAX_JOBS:
            raise RuntimeError("regmean: more than %d solves in one merge" % self.MAX_JOBS)
        self.jobs.append([what, num, den, None, None])
        return num

    def launch(self, small_ready=None):
        """Issue every factorisation and solve (no synchronisation).  The numerators may still be in flight on the stream when this
        is called: the copies kept for the fallback are taken here, in stream order after them (both are overwritten in place).
        small_ready: an event after which every product with a reduction of <= 1 024 is complete (_Products.small_event) -- the
        768-wide groups' side streams wait for IT, not for the whole caller's stream, and start under the 3 072-wide products."""
        if not self.jobs:
            return
        groups = {}
        for i, (what, num, den, _, _) in enumerate(self.jobs):
            groups.setdefault((den.shape[0], num.shape[0], num.stride(0)), []).append(i)
        order = [i for key in sorted(groups) for i in groups[key]]  # status slots in issue order
        self.slot = {j: k for k, j in enumerate(order)}
        # One stream per shape group (round 5): a group's factorisation + solve is a chain of ~300 small dependent launches (a block
        # factor is ONE wave per matrix, a panel solve a few hundred workgroups), and the groups are independent -- side by side the
        # 768-wide groups hide under the 3072-wide one.  The largest group stays on the caller's stream.
        main = torch.cuda.current_stream(self.device)
        keys = sorted(groups, key=lambda k: -k[0] * k[0] * len(groups[k]))
        starts, k0 = {}, 0
        for key in sorted(groups):
            starts[key] = k0
            k0 += len(groups[key])
        side = _side_streams(self.device, len(keys) - 1)
        for n_, key in enumerate(keys):
            idx = groups[key]
            st = main if n_ == 0 else side[n_ - 1]
            if st is not main:
                if small_ready is not None and key[0] <= 1024:
                    st.wait_event(small_ready)
                else:
                    st.wait_stream(main)
            with torch.cuda.stream(st):
                for i in idx:  # (on the group's own stream: before its in-place factorisation, after its products)
                    self.jobs[i][3], self.jobs[i][4] = self.jobs[i][2].clone(), self.jobs[i][1].clone()
                dens = [self.jobs[i][2] for i in idx]
                ops.cholesky_batched_(dens, self.status[starts[key]:starts[key] + len(idx)])
                ops.solve_spd_right_batched_([self.jobs[i][1] for i in idx], dens)
        for st in side[:len(keys) - 1]:
            main.wait_stream(st)

    def collect(self, out):
        """Read the verdicts (ONE synchronisation for all solves) and replace what has no Cholesky factor."""
        if not self.jobs:
            return
        slot = self.slot
        verdict = self.status[:len(self.jobs)].tolist()
        for j, (what, num, den, keep_den, keep_num) in enumerate(self.jobs):
            bad = verdict[slot[j]]
            if bad:
                import warnings
                warnings.warn("regmean: the Gram sum of %s is not positive definite (pivot %d); using a general inverse like "
                              "the reference's torch.inverse" % (what, bad - 1))
                inv = torch.linalg.inv(keep_den).contiguous()
                out[what] = ops.gemm_f64(keep_num, inv, torch.empty_like(num))
        self.jobs = []

    def finish(self, out):
        self.launch()
        self.collect(out)


def _solve(num, den, what):
    """One W* = num @ inverse(den) with the same SPD-or-fallback rule (a single-job _Solves; tests and one-off callers)."""
    out = {}
    s = _Solves(num.device)
    res = s.submit(num, den, what)
    s.finish(out)
    return out.get(what, res)


class _Products:
    """The numerators' terms num (+)= W_m.double() @ G'_m (:421-423), collected per (term, shape) and run `count` at a time in one
    launch (round 5: one product is 144..576 workgroups on 1024 slots).  A group is flushed as soon as it holds eight rounds of
    workgroups, so the device starts while the host is still walking the layers; a weight's term 0 (beta = 0) always precedes its
    term 1 on the stream."""

    ROUNDS = 8 * 1024  # (1 024 / 2 048 / 4 096 / 8 192 tiles: 0.042 / 0.039 / 0.038-0.044 / 0.037-0.038 s per merge)

    def __init__(self):
        self.groups = {}
        self.small_event = None  # recorded after the latest launch of products whose reduction is <= 1 024 long
        self.last_group = {}     # id(num) -> the group that holds the numerator's latest term

    def add(self, term, W, Gs, num):
        key = (tuple(W.shape), W.dtype)
        # A weight's terms are ordered by construction only INSIDE a group.  If its modalities differ in dtype (an fp32 master
        # beside an fp16 tensor widened to float64), term 1 (beta = 1) lands in another group than term 0 (beta = 0), and a
        # later flush of that group could run it first -- term 0 would then overwrite the accumulated numerator.  So the group
        # holding the earlier term is flushed before the later term is queued anywhere else.
        prev = self.last_group.get(id(num))
        if prev is not None and prev != key and prev in self.groups:
            self._flush(prev)
        self.last_group[id(num)] = key
        g = self.groups.setdefault(key, {})
        g.setdefault(term, []).append((W, Gs, num))
        tiles = -(-W.shape[0] // 64) * -(-Gs.shape[1] // 64)
        if tiles * len(g[term]) >= self.ROUNDS:
            self._flush(key, upto=term)

    def _flush(self, key, upto=None):
        g = self.groups[key]
        for term in sorted(g):
            if upto is not None and term > upto:
                break
            if g[term]:
                ws, gs, ns = zip(*g[term])
                ops.gemm_f64_batched(list(ws), list(gs), list(ns), beta=0.0 if term == 0 else 1.0)
                g[term] = []
                if key[0][1] <= 1024:
                    self.small_event = torch.cuda.Event()
                    self.small_event.record()

    def flush(self):
        for key in sorted(self.groups, key=lambda k: k[0][1]):  # short reductions first: their solves may start under the long ones
            self._flush(key)
        self.groups = {}
        self.last_group = {}


def regmean(state_dict, config, gram_matrices=None, device="cuda", plan_out=None):
    passthrough = M._passthrough(state_dict)
    if gram_matrices is None:
        from . import checkpoint
        gram_matrices = checkpoint.load_file(config["gram_matrices"])
    alpha = config["scaling_for_non_diag"]
    plan = M.MergePlan(device)
    dev = plan.device
    solves = _Solves(dev)
    products = _Products()
    merged, order, plain = {}, [], []
    # first the linear weights of every layer (they feed the device: ~150 launches of products, then the solves' ~420), then the
    # plain averages' bookkeeping while those run; `out` is assembled in the reference's key order at the end
    for i in range(M.NUM_MERGE_LAYERS):
        mods = M.modalities_for_layer(config, i, honour_only_used=False)
        for src, dst in M._tensor_names(i):
            order.append(dst)
            is_weight = dst.endswith(".weight") and "norm" not in dst
            if not is_weight:
                plain.append((src, dst, mods))
                continue
            num, den, through, terms = None, None, None, 0
            for m in mods:
                name = src(m)
                gname = name.replace(".qkv.weight", "") if "qkv" in name else name.replace(".weight", "")
                if name in state_dict:
                    if gname not in gram_matrices:
                        continue  # :419-420 (vl experts never get a gram)
                    G = gram_matrices[gname].to(dev, torch.float64).contiguous()
                    Gs = torch.empty_like(G)
                    ops.scale_gram(G, Gs, alpha)                         # G' = a G + (1 - a) diag(G)   (:388-392)
                    W = state_dict[name].to(dev)
                    W = W.contiguous() if W.dtype in (torch.float32, torch.float64) else W.double().contiguous()
                    if num is None:
                        den = Gs
                        num = torch.empty(W.shape[0], G.shape[0], device=dev, dtype=torch.float64)
                    else:
                        den = den + Gs
                    products.add(terms, W, Gs, num)                      # num (+)= W_m.double() @ G'_m   (:421-423)
                    terms += 1
                else:
                    through = state_dict[dst]
                    break
            if through is not None:
                merged[dst] = through
            elif num is None:
                merged[dst] = 0  # the reference's untouched accumulator (no modality had a gram; does not occur in practice)
            else:
                merged[dst] = solves.submit(num, den, dst)
    products.flush()
    solves.launch(small_ready=products.small_event)
    for src, dst, mods in plain:
        srcs, through = M._collect(state_dict, src, dst, mods)
        merged[dst] = through if srcs is None else plan.add(L.MERGE_MEAN, [t for _, t in srcs], None)
    if plan.jobs:
        plan.run()
    solves.collect(merged)
    out = passthrough
    for dst in order:
        out[dst] = merged[dst]
    if plan_out is not None:
        plan_out.append(plan)
    return out
