"""Pin the merge oracle (oracle/merge_oracle.py + merge_ref.c) against what the reference produced."""
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import merge_oracle as mo
from oracle import synth
from oracle.detweights import det_array, det_gram

CASES = {
    "interp_r0.5": ("merge_weights", dict(merge_ratio=0.5)),
    "interp_r0.3": ("merge_weights", dict(merge_ratio=0.3)),
    "interp_r0.3_used_irtr": ("merge_weights", dict(merge_ratio=0.3, only_activate_used_experts=True,
                                                    loss_names={"irtr": 1})),
    "interp_r0.5_used_vqa": ("merge_weights", dict(merge_ratio=0.5, only_activate_used_experts=True,
                                                   loss_names={"vqa": 1})),
    "taskvec_l0.75": ("sum_task_vectors", dict(sum_lambda=0.75)),
    "taskvec_l0.4_used_irtr": ("sum_task_vectors", dict(sum_lambda=0.4, only_activate_used_experts=True,
                                                        loss_names={"irtr": 1})),
    "regmean_a1.0": ("regmean", dict(scaling_for_non_diag=1.0, loss_names={"irtr": 1})),
    "regmean_a0.9": ("regmean", dict(scaling_for_non_diag=0.9, loss_names={"irtr": 1})),
    "regmean_a0.9_pretrain": ("regmean", dict(scaling_for_non_diag=0.9, loss_names={"itm": 1, "mlm": 1, "ifm": 1})),
    "interp_already_ufo": ("merge_weights", dict(merge_ratio=0.5)),
}


def merge_cfg(**over):
    cfg = dict(vlffn_start_layer_index=10, only_activate_used_experts=False, merge_ratio=0.5, sum_lambda=1,
               scaling_for_non_diag=1, loss_names={})
    cfg.update(over)
    return cfg


def tiny_state(arch="all_moe", salt=0, D=16, F=32):
    shapes = synth.state_shapes(D, F, arch, R=24, vocab=32, T=8, heads=2)
    return {k: det_array(k, s, salt) for k, (s, dt) in shapes.items()}


def tiny_grams(D=16, F=32):
    return {k: det_gram(k, s[0]) for k, s in synth.gram_shapes(D, F).items()}


def run_oracle(case):
    fn, over = CASES[case]
    cfg = merge_cfg(**over)
    sd = tiny_state("ufo" if case == "interp_already_ufo" else "all_moe")
    if fn == "merge_weights":
        return mo.merge_weights(sd, cfg), sd
    if fn == "sum_task_vectors":
        return mo.sum_task_vectors(sd, cfg, tiny_state("ufo", salt=7)), sd
    return mo.regmean(sd, cfg, tiny_grams()), sd


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_matches_reference_tiny(case, golden_dir):
    gold = np.load(os.path.join(golden_dir, "merge_tiny.npz"))
    res, sd = run_oracle(case)
    keys = json.loads(str(gold[case + "/__keys__"]))
    assert sorted(res.keys()) == keys
    n = 0
    for k in keys:
        if "transformer.blocks." in k and "gamma" not in k:
            g = gold[case + "/" + k]
            o = np.asarray(res[k])
            assert o.dtype == g.dtype, (k, o.dtype, g.dtype)
            if g.dtype == np.float32:
                assert o.tobytes() == g.tobytes(), k  # bit-exact
            else:  # regmean fp64 weights: LAPACK inverse order of operations differs
                np.testing.assert_allclose(o, g, rtol=1e-9, atol=1e-12)
            n += 1
        else:
            assert res[k] is sd[k]  # pass-through is the same object (checklist item 1)
    assert n == 12 * 13


def test_oracle_negative_zero_and_ragged():
    a = np.array([-0.0, 1.0, -2.5, 3.0, 7.0], dtype=np.float32)
    out = mo.lerp([a], [1])
    assert np.signbit(out[0]) == False  # noqa: E712   0 + (-0.0) = +0.0 as in the reference
    assert out.tobytes()[4:] == a.tobytes()[4:]
    b = np.arange(5, dtype=np.float32)
    tv = mo.taskvec(a, [b, b], [0.75, 0.75])
    c1 = a + np.float32(0.75) * (b - a)
    c2 = c1 + np.float32(0.75) * (b - c1)
    assert tv.tobytes() == c2.tobytes()
    assert mo.mean([a, b, b]).tobytes() == (((np.float32(0) + a) + b + b) / np.float32(3)).tobytes()


def test_oracle_matches_reference_base_digests(golden_dir):
    """Base-size (26 x 7 087 104 params) all_moe -> ufo through the oracle == the reference's sha256."""
    dig = json.load(open(os.path.join(golden_dir, "merge_base_digests.json")))
    shapes = synth.block_shapes(768, 3072, "all_moe")
    # layer 0 (2-way) and layer 11 (3-way) only: keeps the CPU suite short; the GPU test covers all 12
    for layer in (0, 11):
        sd = {k: det_array(k, s) for k, (s, dt) in shapes.items() if k.startswith(f"transformer.blocks.{layer}.")}
        pre = f"transformer.blocks.{layer}."
        for src, dst in mo._names(layer):
            mods = ["v", "l"] if layer < 10 else ["v", "l", "vl"]
            r = 0.3
            ratios = {"v": r, "l": 1 - r} if len(mods) == 2 else {"v": (2 / 3) * r, "l": (2 / 3) * (1 - r), "vl": 1 / 3}
            out = mo.lerp([sd[src(m)] for m in mods], [ratios[m] for m in mods])
            assert hashlib.sha256(out.tobytes()).hexdigest() == dig["interp_r0.3"][dst], dst
            assert dst.startswith(pre)
