"""Gram cache (K15) and RegMean (K14) on the GPU vs the reference's golden outputs / the pinned oracle."""
import importlib
import json
import os

import numpy as np
import pytest
import torch

from oracle import merge_oracle as mo
from oracle.detweights import det_batch
from test_model_gpu import build, gpu_batch
from test_oracle_merge import CASES, merge_cfg, tiny_grams, tiny_state

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(pkg):
    return (importlib.import_module("vl_merging_amd.vilt.config"),
            importlib.import_module("vl_merging_amd.vilt.modules.vilt_module"))


def test_gram_capture_matches_reference(mods, golden_dir):
    """Same 96 keys as the reference's hook produces.  The products and sums are float64 on the device (v_mfma_f64
    SYRK); what differs from the reference is the ACTIVATIONS the engine computed upstream in bf16 (and, for the
    attention output and the GELU output, the bf16 tensors themselves): relative Frobenius error of a Gram <= 5e-3."""
    gold = np.load(os.path.join(golden_dir, "irtr_tiny_all_moe.npz"))
    model = build(mods, "all_moe", "tiny_irtr_all_moe", golden_dir, {"irtr": 1, "itm": 0, "mlm": 0, "ifm": 0}, max_vl=None)
    batch = gpu_batch(det_batch(3, 224, 40, 1024, seed=77))
    mods[1].vilt_utils.set_task(model)
    cap = model.start_gram_capture()
    with torch.no_grad():
        model(batch)
    model.stop_gram_capture()
    grams = cap.state_dict()
    assert sorted(grams) == json.loads(str(gold["gram_keys"]))
    summ = json.loads(str(gold["gram_summary"]))
    for k, (shape, nrm, sm) in summ.items():
        g = grams[k]
        assert g.dtype == torch.float64 and list(g.shape) == shape
        assert abs(float(g.norm()) - nrm) <= 5e-3 * nrm, (k, float(g.norm()), nrm)
    for key in gold.files:
        if key.startswith("gram/"):
            g = grams[key[5:]][:192, :192].numpy()
            ref = gold[key]
            assert np.linalg.norm(g - ref) <= 5e-3 * np.linalg.norm(ref), key
    # a second identical batch doubles every accumulator (accumulation across hook calls)
    cap2 = model.start_gram_capture()
    with torch.no_grad():
        model(batch)
        model(batch)
    model.stop_gram_capture()
    k0 = sorted(grams)[0]
    assert torch.allclose(cap2.grams[k0].cpu(), 2 * grams[k0], rtol=1e-6, atol=1e-9)


def test_gram_capture_base_width_matches_reference(mods, golden_dir):
    """configs[3]'s capture leg at the BASE geometry (hidden 768, F 3072, 384^2, N = 617, all_moe irtr model) through
    engine.GramCapture, against the reference's own hook on the same deterministic weights and batch
    (tests/golden/gram_base.npz, make_golden.py gram_base; reference cache_gram_matrices.py:246-281).
    Tolerance 1e-2 relative Frobenius per matrix, stated next to what reduced precision does to a Gram: the products and
    sums are float64 here as in the reference; the INPUTS are the engine's activations (bf16 GEMM operands: 8-bit
    mantissa, rel. rounding 2^-9 = 2e-3 per element, compounding over up to 12 layers; the reference's fp16-AMP features
    deviate 1.7e-3 of the scale at this width, amp_reference_errors.json).  Measured values go to parity_errors.json."""
    from test_model_gpu import build_base, MEASURED
    gold = np.load(os.path.join(golden_dir, "gram_base.npz"))
    model = build_base(mods, "all_moe", golden_dir, {"irtr": 1, "itm": 0, "mlm": 0, "ifm": 0}, tag="base_irtr_all_moe", max_vl=None)
    batch = gpu_batch(det_batch(3, 384, 40, 1024, seed=99))
    mods[1].vilt_utils.set_task(model)
    cap = model.start_gram_capture()
    with torch.no_grad():
        model(batch)
    model.stop_gram_capture()
    grams = cap.state_dict()
    assert sorted(grams) == json.loads(str(gold["gram_keys"])) and len(grams) == 96
    summ = json.loads(str(gold["gram_summary"]))
    worst_norm, worst_block = 0.0, 0.0
    for k, (shape, nrm, sm) in summ.items():
        g = grams[k]
        assert g.dtype == torch.float64 and list(g.shape) == shape, k
        rel = abs(float(g.norm()) - nrm) / nrm
        worst_norm = max(worst_norm, rel)
        assert rel <= 1e-2, (k, float(g.norm()), nrm)
        assert float((g - g.t()).abs().max()) <= 1e-12 * float(g.abs().max()), k  # mirrored upper triangle (float64 atomics: order, not value)
    picked = [key for key in gold.files if key.startswith("gram/")]
    assert len(picked) == 8
    for key in picked:
        g = grams[key[5:]][:256, :256].numpy()
        ref = gold[key].astype(np.float64)
        rel = np.linalg.norm(g - ref) / np.linalg.norm(ref)
        worst_block = max(worst_block, float(rel))
        assert rel <= 1e-2, (key, rel)
    MEASURED.setdefault("test_gram_capture_base_width_matches_reference", {}).update(
        {"gram_norm_rel_max": worst_norm, "gram_256_block_rel_frobenius_max": worst_block})


@pytest.mark.parametrize("case", [c for c in sorted(CASES) if CASES[c][0] == "regmean"])
def test_regmean_matches_reference_tiny(case, pkg, golden_dir):
    rm = importlib.import_module("vl_merging_amd.regmean")
    gold = np.load(os.path.join(golden_dir, "merge_tiny.npz"))
    fn, over = CASES[case]
    cfg = merge_cfg(**over)
    sd = {k: torch.from_numpy(v).cuda() for k, v in tiny_state("all_moe").items()}
    grams = {k: torch.from_numpy(v) for k, v in tiny_grams().items()}
    res = rm.regmean(sd, cfg, gram_matrices=grams)
    torch.cuda.synchronize()
    keys = json.loads(str(gold[case + "/__keys__"]))
    assert sorted(res.keys()) == keys
    for k in keys:
        if "transformer.blocks." in k and "gamma" not in k:
            g = gold[case + "/" + k]
            o = res[k].cpu().numpy()
            assert o.dtype == g.dtype, (k, o.dtype, g.dtype)
            if g.dtype == np.float32:
                assert o.tobytes() == g.tobytes(), k       # averages: HIP merge kernel, bit-exact
            else:
                np.testing.assert_allclose(o, g, rtol=1e-8, atol=1e-11)  # fp64 GEMM + inverse on device
        else:
            assert res[k] is sd[k]


def test_regmean_with_modalities_of_different_dtypes(pkg, golden_dir):
    """A weight whose two modalities arrive in different dtypes (fp32 beside fp16, which the merge widens to float64): its two
    numerator terms land in different product batches.  Term 0 (beta = 0) must still run before term 1 (beta = 1); the result
    is the tiny golden's up to the fp16 rounding of the one modality, i.e. equal to the merge of the ROUNDED state in fp32."""
    rm = importlib.import_module("vl_merging_amd.regmean")
    case = [c for c in sorted(CASES) if CASES[c][0] == "regmean"][0]
    cfg = merge_cfg(**CASES[case][1])
    base = tiny_state("all_moe")
    grams = {k: torch.from_numpy(v) for k, v in tiny_grams().items()}
    halved = lambda k: ".l." in k and k.endswith(".weight") and "norm" not in k  # the language experts' linear weights
    sd_mixed = {k: (torch.from_numpy(v).cuda().half() if halved(k) else torch.from_numpy(v).cuda()) for k, v in base.items()}
    sd_round = {k: (torch.from_numpy(v).cuda().half().float() if halved(k) else torch.from_numpy(v).cuda()) for k, v in base.items()}
    assert any(halved(k) for k in base)
    got = rm.regmean(sd_mixed, cfg, gram_matrices=grams)
    want = rm.regmean(sd_round, cfg, gram_matrices=grams)
    torch.cuda.synchronize()
    n = 0
    for k, w in want.items():
        if torch.is_tensor(w) and w.dtype == torch.float64:
            np.testing.assert_allclose(got[k].cpu().numpy(), w.cpu().numpy(), rtol=1e-10, atol=1e-12, err_msg=k)
            n += 1
    assert n > 0


def test_regmean_base_size_matches_reference(pkg, golden_dir):
    """configs[3], RegMean half, at base size (D = 768, F = 3072): layers 0 and 11 of an all_moe state through the
    MFMA-f64 GEMMs + blocked Cholesky solve against the reference's `W.double() @ G` / torch.inverse results
    (tests/golden/regmean_base.npz: norms, three rows and the column sums of every merged weight).  1e-8 relative."""
    import sys
    sys.path.insert(0, golden_dir)
    rm = importlib.import_module("vl_merging_amd.regmean")
    gold = np.load(os.path.join(golden_dir, "regmean_base.npz"))
    from oracle import synth
    from oracle.detweights import det_array, det_gram
    layers = (0, 11)
    D, F = 768, 3072
    sd = {}
    for k, (shp, dt) in synth.block_shapes(D, F, "all_moe").items():
        if int(k.split(".")[2]) in layers or "gamma" in k:
            sd[k] = torch.from_numpy(det_array(k, shp)).cuda()
    for k, (shp, dt) in synth.block_shapes(D, F, "ufo").items():
        if int(k.split(".")[2]) not in layers and "gamma" not in k:
            sd[k] = torch.from_numpy(det_array(k, shp, 5)).cuda()
    grams = {k: torch.from_numpy(det_gram(k, s[0])) for k, s in synth.gram_shapes(D, F).items() if int(k.split(".")[2]) in layers}
    cfg = merge_cfg(scaling_for_non_diag=0.9, loss_names={"irtr": 1})
    res = rm.regmean(sd, cfg, gram_matrices=grams)
    torch.cuda.synchronize()
    checked = 0
    for key in gold.files:
        if key.endswith("/norm"):
            k = key[:-5]
            a = res[k]
            assert a.dtype == torch.float64
            want = float(gold[key])
            assert abs(float(a.norm()) - want) <= 1e-9 * want, k
            rows = a[[0, a.shape[0] // 2, a.shape[0] - 1]][:, :256].cpu().numpy()
            np.testing.assert_allclose(rows, gold[k + "/rows"], rtol=1e-8, atol=1e-10 * want)
            np.testing.assert_allclose(a.sum(0)[:256].cpu().numpy(), gold[k + "/colsum"], rtol=1e-8, atol=1e-9 * want)
            checked += 1
    assert checked == 8  # qkv, proj, fc1, fc2 of two layers


def test_engine_grams_feed_regmean_like_reference_grams(mods, golden_dir):
    """End to end (tiny width): Gram matrices CAPTURED BY THE ENGINE on an irtr batch -> regmean, against the oracle's
    float64 Gram matrices of the same batch (pinned to the reference's hook outputs, tests/test_oracle_model.py) ->
    the same regmean.  The merged weights differ only through the engine's bf16 activations: <= 2e-2 of the weight
    scale (RegMean divides by the sum of Grams, so their relative error carries over once)."""
    from oracle import vlmo_ref as R
    rm = importlib.import_module("vl_merging_amd.regmean")
    model = build(mods, "all_moe", "tiny_irtr_all_moe", golden_dir, {"irtr": 1, "itm": 0, "mlm": 0, "ifm": 0}, max_vl=None)
    nb = det_batch(3, 224, 40, 1024, seed=77)
    batch = gpu_batch(nb)
    mods[1].vilt_utils.set_task(model)
    cap = model.start_gram_capture()
    with torch.no_grad():
        model(batch)
    model.stop_gram_capture()
    eng = cap.state_dict()
    sd_cpu = {k: v.detach().float().cpu() for k, v in model.state_dict().items() if v.is_floating_point()}
    idx = {k: getattr(model, k).cpu() for k in ("relative_position_index", "text_relative_position_index",
                                                 "text_imag_relative_position_index")}
    with torch.no_grad():
        ref = R.gram_inputs(sd_cpu, R.Arch("all_moe", hidden=192, heads=3), idx, {k: torch.from_numpy(v) for k, v in nb.items()})
    assert sorted(ref) == sorted(eng)
    # ridge the tiny-batch Grams (3 x 237 tokens < 768 fc2 inputs: singular without it) the same way on both sides
    def ridged(gs):
        return {k: v.double() + 1e-3 * float(v.diagonal().mean()) * torch.eye(v.shape[0], dtype=torch.float64) for k, v in gs.items()}
    cfg = merge_cfg(scaling_for_non_diag=0.9, loss_names={"irtr": 1})
    sd = {k: v for k, v in model.state_dict().items() if "transformer.blocks." in k}
    a = rm.regmean(dict(sd), cfg, gram_matrices=ridged(eng))
    b = rm.regmean(dict(sd), cfg, gram_matrices=ridged(ref))
    torch.cuda.synchronize()
    n = 0
    for k, v in a.items():
        if torch.is_tensor(v) and v.dtype == torch.float64:
            err = float((v - b[k]).abs().max())
            scale = float(b[k].abs().max())
            assert err <= 2e-2 * scale, (k, err, scale)
            n += 1
    assert n == 48


def test_regmean_non_positive_definite_gram_falls_back_like_the_reference(pkg):
    """scaling_for_non_diag = 1 with a Gram sum that has no Cholesky factor (numerically indefinite: a Gram of few capture
    batches).  The reference's torch.inverse (LU) still returns a result (vilt_module.py:432); the device path must not
    abort the merge: it warns and uses a general float64 inverse -- same W* as num @ inverse(den) in numpy float64."""
    import warnings
    rm = importlib.import_module("vl_merging_amd.regmean")
    g = torch.Generator().manual_seed(5)
    D, O = 128, 96
    A = torch.randn(D, D, generator=g, dtype=torch.float64)
    spd = A @ A.t() / D + 0.5 * torch.eye(D, dtype=torch.float64)
    evals, evecs = torch.linalg.eigh(spd)
    evals[0] = -0.05                                   # one slightly negative eigenvalue: invertible, not positive definite
    den = (evecs * evals) @ evecs.t()
    den = 0.5 * (den + den.t())
    num = torch.randn(O, D, generator=g, dtype=torch.float64)
    want = num.numpy() @ np.linalg.inv(den.numpy())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        got = rm._solve(num.cuda().contiguous(), den.cuda().contiguous(), "test.weight")
    assert any("not positive definite" in str(x.message) for x in w)
    np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-8, atol=1e-9)
    # a positive definite sum takes the Cholesky path silently and agrees too
    with warnings.catch_warnings(record=True) as w2:
        warnings.simplefilter("always")
        got2 = rm._solve(num.cuda().contiguous(), spd.cuda().contiguous(), "test.weight")
    assert not w2
    np.testing.assert_allclose(got2.cpu().numpy(), num.numpy() @ np.linalg.inv(spd.numpy()), rtol=1e-8, atol=1e-9)
