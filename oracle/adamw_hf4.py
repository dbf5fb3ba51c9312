"""float64 restatement of the optimizer the reference trains with: transformers 4.x `AdamW` (decoupled weight decay AFTER the
Adam update, bias correction on) under `get_polynomial_decay_schedule_with_warmup`.  TEST INFRASTRUCTURE ONLY (tests/,
__graft_entry__.smoke(), bench.py's cpu_baseline leg): the product's optimizer is vl-merging_amd/csrc/elementwise.hip
`adamw_kernel` behind vilt_utils.FusedAdamW.

Reference call sites: src/vilt/modules/vilt_utils.py:314-317 (`AdamW(optimizer_grouped_parameters, lr=lr, eps=1e-8,
betas=(0.9, beta_2))`), :330-352 (the schedule), :272-312 (the four parameter groups).

parity: UNPINNED.  The class was removed from the installed transformers (5.x) and its source is not under /root/reference, so
the rule below is the PUBLISHED 4.x one (optimization.py, `AdamW.step`), restated:
    exp_avg    <- b1 exp_avg + (1 - b1) g
    exp_avg_sq <- b2 exp_avg_sq + (1 - b2) g g
    step_size   = lr sqrt(1 - b2^t) / (1 - b1^t)           (correct_bias=True; t counts the parameter's own updates)
    p          <- p - step_size exp_avg / (sqrt(exp_avg_sq) + eps)
    p          <- p - lr weight_decay p                     (after the update, with the decayed lr)
and a parameter whose .grad is None is skipped entirely (no decay, no state).  What IS pinned around it: the parameter groups,
learning rates and the schedule (tests/golden/schedule_groups.json, produced by the reference's own set_schedule).
"""
import math

import numpy as np


def polynomial_decay_with_warmup(step, warmup, total, lr_init, lr_end, power):
    """lr FACTOR of transformers' get_polynomial_decay_schedule_with_warmup (the LambdaLR's lambda)."""
    if step < warmup:
        return float(step) / float(max(1, warmup))
    if step > total:
        return lr_end / lr_init
    lr_range = lr_init - lr_end
    decay_steps = total - warmup
    pct_remaining = 1 - (step - warmup) / decay_steps
    return (lr_range * pct_remaining ** power + lr_end) / lr_init


class AdamWHF4:
    """params: name -> float64 ndarray (updated in place); groups: name -> (initial_lr, weight_decay)."""

    def __init__(self, params, groups, betas=(0.9, 0.999), eps=1e-8):
        self.p, self.groups, self.b1, self.b2, self.eps = params, groups, betas[0], betas[1], eps
        self.state = {}

    def step(self, grads, lr_factor):
        for n, g in grads.items():
            if g is None:
                continue  # `if p.grad is None: continue`
            g = np.asarray(g, dtype=np.float64)
            st = self.state.setdefault(n, {"t": 0, "m": np.zeros_like(self.p[n]), "v": np.zeros_like(self.p[n])})
            st["t"] += 1
            st["m"] = self.b1 * st["m"] + (1 - self.b1) * g
            st["v"] = self.b2 * st["v"] + (1 - self.b2) * g * g
            lr0, wd = self.groups[n]
            lr = lr0 * lr_factor
            step_size = lr * math.sqrt(1 - self.b2 ** st["t"]) / (1 - self.b1 ** st["t"])
            self.p[n] -= step_size * st["m"] / (np.sqrt(st["v"]) + self.eps)
            if wd > 0.0:
                self.p[n] -= lr * wd * self.p[n]
