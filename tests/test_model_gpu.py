"""The MI355X model (HIP kernels, bf16 GEMMs / fp32 accumulation) against the reference's golden outputs and the
fp32 oracle on the same deterministic weights and batch.

Stated tolerance (north star: "fp within a stated tol for attention/FFN"): activations are rounded to bf16 between
kernels (8 significant bits, like the reference's fp16 AMP path rounds to 11), so
   features after 12 blocks : max |err| <= 1.7e-2 * max|ref| (1.5 x the measured 1.14e-2 at base width; <= 0.7e-2 at tiny width)
   logits                   : max |err| <= 2e-2 * max|ref|   (1.5 x the measured 1.36e-2; the B = 3 tiny-width contrastive logits: 4.5e-2)
   losses                   : |err| <= 3e-2 (absolute, values are O(1..10))
   parameter-gradient norms : relative error <= 2.5e-2 per tensor (1.5 x the measured 1.5e-2, train mode all_moe; bf16 wgrad operands)
"""
import importlib
import json
import os

import numpy as np
import pytest
import torch

from oracle.detweights import det_array, det_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods(pkg):
    return (importlib.import_module("vl_merging_amd.vilt.config"),
            importlib.import_module("vl_merging_amd.vilt.modules.vilt_module"))


def build(mods, arch, tag, golden_dir, losses, max_vl=40, train=False):
    cfgmod, vm = mods
    cfg = cfgmod.make_config(arch, vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, vocab_size=1024,
                             max_text_len=40, patch_size=16, vlffn_start_layer_index=10, image_size=224,
                             max_vl_text_len=max_vl, tasks=["vl"] if max_vl else None,
                             loss_names=cfgmod._loss_names(losses))
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    meta = json.load(open(os.path.join(golden_dir, f"keys_{tag}.json")))
    sd = {k: torch.from_numpy(det_array(k, s)) for k, (s, dt) in meta.items()
          if dt.startswith("float") and "index" not in k and "mask_for" not in k and not k.startswith(("train_", "val_"))}
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    model = model.cuda()
    model.train(train)
    model.setup_engine()
    return model


def gpu_batch(nb):
    b = {k: torch.from_numpy(v).cuda() for k, v in nb.items()}
    b["image"] = [b["image"]]
    return b


FLOOR = 2.5e-4  # absolute gradient-norm floor: below it ANY reduced-precision path is noise (see grad_norm_ok)


def grad_norm_class(got, ref):
    """Which rule lets a tensor's gradient norm pass: "rel" (2.5 %), "small" (norm <= 0.05: 20 %), "floor" (absolute
    2.5e-4 only), or None (fails)."""
    rel = abs(got - ref) / (ref + 1e-12)
    if rel <= 2.5e-2:
        return "rel"
    if ref <= 0.05 and rel <= 0.20:
        return "small"
    if abs(got - ref) <= FLOOR:
        return "floor"
    return None


def grad_norm_ok(got, ref):
    """2.5 % per tensor (measured: max 1.5 %, median 0.1 %).  Gradients whose norm is below 2.5e-4 in
    absolute terms (logit_vl_scale at B = 2: 3.9e-5) are below the noise floor of ANY reduced-precision path: the
    reference's own fp16-autocast run moves that one by 114 % (tests/golden/amp_reference_errors.json).
    Tensors with a small gradient (norm <= 0.05: here those fed only by the B=2 contrastive
    losses, i.e. by the DIFFERENCE of two nearly identical L2-normalised features scaled by exp(logit_scale) ~ 14,
    ill-conditioned in any 8-bit-mantissa activation format, and run-to-run sensitive to the order of the fp32 atomic
    accumulations) get 20 %; near-zero scalar gradients an absolute 1e-4."""
    rel = abs(got - ref) / (ref + 1e-12)
    return rel <= 2.5e-2 or (ref <= 0.05 and rel <= 0.20) or abs(got - ref) <= 2.5e-4


def feat_close(got, ref, what, tol=1.7e-2):
    got = got.float().cpu()
    ref = torch.as_tensor(ref)
    err = float((got - ref).abs().max())
    scale = float(ref.abs().max())
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0]
    MEASURED.setdefault(test, {})[what] = err / (scale + 1e-30)
    assert err <= tol * scale, "%s: max err %.4g vs scale %.4g" % (what, err, scale)


def roundoff_model_bound(n_elements, roundings=144, cancellation=1.5):
    """Expected max over a tensor's elements of |gradient error| / max |gradient| under the bf16 round-off model, as a bound.
    Every activation and every gradient is rounded to bf16 where it crosses a kernel boundary: a relative error uniform in
    +-2^-9 (rms 2^-9 / sqrt 3 = 1.13e-3).  On the longest path of a training step a value crosses about 12 such boundaries per
    block (forward: LayerNorm output, qkv, attention output, proj input, LayerNorm output, fc1 / GELU output; backward: the
    mirror images) x 12 blocks = 144 independent roundings, which add as a random walk: sigma = sqrt(144) x 1.13e-3 = 1.35 % of
    the signal.  The largest of N such errors is about sqrt(2 ln N) sigma (4.6 sigma for a 192 x 192 weight, 3.2 for a 192-vector),
    and an element that is a sum of partly cancelling terms sees sigma times (sum |terms| / |sum|): `cancellation`, taken as 1.5.
    => 192-vector 6.6 %, 192 x 192 weight 9.3 %, 768 x 768 weight 10.4 %.  (What the final GEMM's own operand rounding adds is
    sqrt(K) 2^-8 of ONE product against a sum of K: two orders below this.)  For comparison, the reference's own fp16-autocast
    run moves the worst sampled tensor by 6.3 % (ufo) on the same check at base width (amp_reference_errors.json)."""
    import math
    sigma = math.sqrt(roundings) * 2.0 ** -9 / math.sqrt(3.0)
    return cancellation * math.sqrt(2.0 * math.log(max(n_elements, 2))) * sigma


def grad_elementwise_close(got, ref, name, tol_small=0.2, tol=None):
    """Element-wise check of a sampled gradient tensor against the reference's: max |got - ref| relative to the tensor's
    largest entry.  The bound is DERIVED (roundoff_model_bound: 6.6-10.4 % by tensor size), not "2 x measured" as in round 5;
    measured (recorded into parity_errors.json by every run): 6.8-9.6 % at the tiny width -- the worst tensor is cls_token, a sum
    of a few small rows, the one case that needs the small-gradient allowance below -- and 3.6-4.8 % at base width, the same with
    the LayerScale folded or not.  The gradient NORMS, which carry the weight of the parity claim, are checked at 2.5 %."""
    if tol is None:
        tol = roundoff_model_bound(ref.numel())
    err = float((got - ref).abs().max())
    mx = float(ref.abs().max())
    floor = 2.5e-4 if ref.numel() == 1 else 1e-6  # near-zero scalar (logit scale) gradients: see grad_norm_ok
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0]
    rec = MEASURED.setdefault(test, {})
    rec["grad elementwise max rel"] = max(rec.get("grad elementwise max rel", 0.0), max(0.0, err - floor) / (mx + 1e-30))
    assert err <= (tol_small if mx <= 0.05 else tol) * mx + floor, (name, err, mx)


@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_infer_matches_reference_golden(mods, golden_dir, arch):
    gold = np.load(os.path.join(golden_dir, f"model_tiny_{arch}.npz"))
    model = build(mods, arch, f"tiny_{arch}", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1})
    batch = gpu_batch(det_batch(2, 224, 40, 1024, seed=1234))
    with torch.no_grad():
        r = model.infer(batch, mask_text=False)
        for k in ("text_feats", "image_feats", "cls_feats", "raw_cls_feats"):
            feat_close(r[k], gold["infer/" + k], k)
        r = model.infer(batch, mask_text=True)
        feat_close(r["text_feats"], gold["infer_mlm/text_feats"], "mlm text_feats")
        r = model.infer_image(batch)
        for k in ("image_feats", "cls_feats", "cls_vlffn_feats"):
            feat_close(r[k], gold["infer_image/" + k], "image " + k)
        r = model.infer_text(batch)
        for k in ("text_feats", "cls_feats", "cls_vlffn_feats"):
            feat_close(r[k], gold["infer_text/" + k], "text " + k)
        # operator-level entry: Block.forward on a fixed hidden state (reference-shaped call)
        x = (torch.from_numpy(det_array("probe.x", (2, 237, 192))) * 10).cuda()
        mask = torch.cat([batch["text_masks"], torch.ones(2, 197, dtype=torch.long, device="cuda")], 1)
        rp = model.get_rel_pos_bias(model.text_imag_relative_position_index)
        for li in (0, 11):
            y, _ = model.transformer.blocks[li](x, mask=mask, type_id=2, relative_position_bias=rp)
            feat_close(y, gold[f"block{li}/joint"], f"block{li}", tol=1e-2)


@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_training_step_matches_reference_golden(mods, golden_dir, arch):
    gold = np.load(os.path.join(golden_dir, f"model_tiny_{arch}.npz"))
    model = build(mods, arch, f"tiny_{arch}", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1})  # eval mode = golden's mode
    batch = gpu_batch(det_batch(2, 224, 40, 1024, seed=1234))
    model.zero_grad()
    mods[1].vilt_utils.set_task(model)
    ret = model({"vl": batch})
    total = sum(v for k, v in ret.items() if "loss" in k)
    total.backward()
    torch.cuda.synchronize()
    for k in ("mlm_loss", "ifm_loss", "itm_loss"):
        assert abs(float(ret[k]) - float(gold["step/" + k])) <= 3e-2, (k, float(ret[k]), float(gold["step/" + k]))
    assert abs(float(total) - float(gold["step/total_loss"])) <= 5e-2
    feat_close(ret["mlm_logits"], gold["step/mlm_logits"], "mlm logits", tol=2e-2)
    feat_close(ret["itm_logits"], gold["step/itm_logits"], "itm logits", tol=2e-2)
    gs = json.loads(str(gold["step/grad_summary"]))
    named = dict(model.named_parameters())
    bad = []
    for n, v in gs.items():
        g = named[n].grad
        if v is None:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        nrm = float(g.double().norm())
        if not grad_norm_ok(nrm, v[0]):
            bad.append((n, nrm, v[0]))
    assert not bad, bad[:10]
    for key in gold.files:
        if key.startswith("step/grad/"):
            n = key[len("step/grad/"):]
            ref = torch.from_numpy(gold[key])
            got = named[n].grad.float().cpu()
            grad_elementwise_close(got, ref, n)


@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_irtr_matches_reference_golden(mods, golden_dir, arch):
    gold = np.load(os.path.join(golden_dir, f"irtr_tiny_{arch}.npz"))
    model = build(mods, arch, f"tiny_irtr_{arch}", golden_dir, {"irtr": 1, "itm": 0, "mlm": 0, "ifm": 0}, max_vl=None)
    batch = gpu_batch(det_batch(3, 224, 40, 1024, seed=77))
    model.zero_grad()
    mods[1].vilt_utils.set_task(model)
    ret = model(batch)
    ret["irtr_loss"].backward()
    assert abs(float(ret["irtr_loss"]) - float(gold["irtr_loss"])) <= 2e-2
    # B = 3 contrastive logits at tiny width: exp(logit_scale) ~ 14 times the DIFFERENCE of nearly identical normalised features
    # (measured 3.0e-2 for ufo, 1.0e-2 for all_moe and at base width, where the bound is 2e-2)
    feat_close(ret["irtr_i2t_logits"], gold["irtr_i2t_logits"], "irtr logits", tol=4.5e-2)
    gs = json.loads(str(gold["grad_summary"]))
    named = dict(model.named_parameters())
    bad = [(n, float(named[n].grad.double().norm()), v[0]) for n, v in gs.items()
           if v is not None and not grad_norm_ok(float(named[n].grad.double().norm()), v[0])]
    assert not bad, bad[:10]


def test_train_mode_step_and_optimizer(mods, golden_dir):
    """Train mode (DropPath + dropout live), two fused AdamW steps: loss finite, weights and bf16 shadows move."""
    model = build(mods, "all_moe", "tiny_all_moe", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1}, train=True)
    batch = gpu_batch(det_batch(4, 224, 40, 1024, seed=5))
    (opt,), (sch,) = model.configure_optimizers() if False else mods[1].vilt_utils.set_schedule(model, max_steps=100)
    w0 = model.transformer.blocks[0].attn["v"].qkv.weight.detach().clone()
    losses = []
    for it in range(2):
        loss = model.training_step({"vl": batch})
        loss.backward()
        opt.step()
        sch["scheduler"].step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)), losses
    w1 = model.transformer.blocks[0].attn["v"].qkv.weight.detach()
    assert float((w1 - w0).abs().max()) > 0
    assert torch.equal(model._flat.flat_b[:model._flat.numel], model._flat.shadow_reference())
    assert float(model._flat.flat_g.abs().max()) == 0.0  # fused zero_grad


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_irtr_recall_matches_reference_golden(mods, golden_dir, arch):
    """compute_irtr_recall end to end on the engine: the two feature sweeps against the reference's features (bf16 GEMM
    tolerance 3e-2 on L2-normalised features) and the six recalls against the reference's own compute_irtr_recall on
    the same 10 images x 3 captions (a recall may move by one query, 0.1 / 0.034, when a bf16-sized score gap flips)."""
    from oracle.detweights import det_batch
    gold = np.load(os.path.join(golden_dir, "irtr_recall_tiny.npz"))
    obj = importlib.import_module("vl_merging_amd.vilt.modules.objectives")
    model = build(mods, arch, f"tiny_irtr_{arch}", golden_dir, {"irtr": 1, "itm": 0, "mlm": 0, "ifm": 0}, max_vl=None)
    ib = det_batch(10, 224, 40, 1024, seed=91)
    tb = det_batch(30, 224, 40, 1024, seed=92)
    texts, images = [], []
    for lo in range(0, 30, 8):  # ragged batches (8, 8, 8, 6): batching must not change the result
        sl = slice(lo, min(30, lo + 8))
        texts.append({"text_ids": torch.from_numpy(tb["text_ids"][sl]), "text_masks": torch.from_numpy(tb["text_masks"][sl]),
                      "text_labels": torch.from_numpy(tb["text_labels"][sl]), "img_index": [j // 3 for j in range(sl.start, sl.stop)]})
    for lo in range(0, 10, 4):
        sl = slice(lo, min(10, lo + 4))
        images.append({"image": [torch.from_numpy(ib["image"][sl])], "img_index": list(range(sl.start, sl.stop)),
                       "text_masks": torch.from_numpy(tb["text_masks"][:1])})
    out = obj.compute_irtr_recall(model, texts, images)
    feats = out[6]
    for name in ("txt_cls_feats", "img_cls_feats"):
        err = float((feats[name].cpu() - torch.from_numpy(gold[f"{arch}/{name}"])).abs().max())
        assert err <= 3e-2, (name, err)
    got = np.array([float(x) for x in out[:6]])
    want = gold[f"{arch}/recalls"]
    step = np.array([1 / 30] * 3 + [1 / 10] * 3)
    assert np.all(np.abs(got - want) <= step + 1e-6), (got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("task", ["vqa", "nlvr2"])
def test_downstream_heads_match_reference_golden(mods, golden_dir, task):
    """VQA (soft-target BCE over the answer vocabulary) and NLVR2 (two joint passes with image token types 1 / 2, pair
    classifier) against the reference's loss, logits and per-parameter gradient norms on deterministic weights."""
    cfgmod, vm = mods
    gold = np.load(os.path.join(golden_dir, "downstream_tiny_ufo.npz"))
    cfg = cfgmod.make_config("ufo", vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, vocab_size=1024,
                             max_text_len=40, patch_size=16, vlffn_start_layer_index=10, image_size=224,
                             vqav2_label_size=37, loss_names=cfgmod._loss_names({task: 1}))
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    meta = json.load(open(os.path.join(golden_dir, f"keys_tiny_{task}_ufo.json")))
    sd = {k: torch.from_numpy(det_array(k, s)) for k, (s, dt) in meta.items()
          if dt.startswith("float") and "index" not in k and "mask_for" not in k and not k.startswith(("train_", "val_", "dev_", "test_"))}
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    assert not [m for m in res.missing_keys if "index" not in m and "position_ids" not in m and "mask_for" not in m], res.missing_keys
    model = model.cuda().eval()
    model.setup_engine()
    nb, nb2 = det_batch(3, 224, 40, 1024, seed=55), det_batch(3, 224, 40, 1024, seed=56)
    if task == "vqa":
        batch = gpu_batch(nb)
        batch["vqa_labels"] = [[3, 17], [0], [5, 6, 30]]
        batch["vqa_scores"] = [[1.0, 0.3], [0.6], [0.9, 0.3, 0.3]]
    else:
        batch = {k: torch.from_numpy(v).cuda() for k, v in nb.items() if k != "image"}
        batch["image_0"] = [torch.from_numpy(nb["image"]).cuda()]
        batch["image_1"] = [torch.from_numpy(nb2["image"]).cuda()]
        batch["answers"] = [1, 0, 1]
    mods[1].vilt_utils.set_task(model)
    model.zero_grad()
    ret = model(batch)
    loss = ret[task + "_loss"]
    loss.backward()
    torch.cuda.synchronize()
    want = float(gold[task + "/loss"])
    assert abs(float(loss) - want) <= 2e-2 * max(1.0, abs(want)), (float(loss), want)
    feat_close(ret[task + "_logits"], gold[task + "/logits"], task + " logits", tol=1e-2)
    gs = json.loads(str(gold[task + "/grad_summary"]))
    named = dict(model.named_parameters())
    bad = []
    for n, v in gs.items():
        g = named[n].grad
        if v is None:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        nrm = float(g.double().norm())
        if not grad_norm_ok(nrm, v[0]):
            bad.append((n, nrm, v[0]))
    assert not bad, bad[:10]


@pytest.mark.gpu
@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_unimodal_pair_pass_equals_separate_passes(mods, golden_dir, arch):
    """infer_unimodal_pair (image-only + text-only pass as one block-diagonal pass) against infer_image + infer_text on
    the same model and batch (eval mode): every row runs the same kernels on the same values, only the launch is shared,
    so the features agree to bf16 rounding of the differently tiled attention (1e-2 of the feature scale)."""
    model = build(mods, arch, f"tiny_{arch}", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1})
    batch = gpu_batch(det_batch(3, 224, 40, 1024, seed=321))
    with torch.no_grad():
        ri, rt = model.infer_image(batch), model.infer_text(batch)
        pi, pt = model.infer_unimodal_pair(batch, with_vlffn=True)
        for k in ("image_feats", "cls_feats", "cls_vlffn_feats", "raw_cls_feats"):
            feat_close(pi[k], ri[k].float().cpu(), "pair image " + k, tol=1e-2)
        for k in ("text_feats", "cls_feats", "cls_vlffn_feats", "raw_cls_feats"):
            feat_close(pt[k], rt[k].float().cpu(), "pair text " + k, tol=1e-2)
        fi, ft = model.infer_unimodal_pair(batch, with_vlffn=False)
        assert fi["cls_vlffn_feats"] is None and ft["cls_vlffn_feats"] is None
        feat_close(fi["cls_feats"], model.infer_image_ft(batch)["cls_feats"].float().cpu(), "pair ft image cls", tol=1e-2)


@pytest.mark.gpu
def test_pair_pass_droppath_draws_are_independent(mods):
    engine = importlib.import_module("vl_merging_amd.engine")
    ops = importlib.import_module("vl_merging_amd.ops")
    pc = engine.PassCtx(ops.Seq(64, 3, 5), 2, None)
    pc.independent_segments = True
    torch.manual_seed(0)
    rs = pc.drop_path_rows(0.5, True, torch.device("cuda"))
    t = rs[: 64 * 3].view(64, 3)
    i = rs[64 * 3:].view(64, 5)
    assert bool((t == t[:, :1]).all()) and bool((i == i[:, :1]).all())       # one draw per sample and segment
    assert set(rs.unique().tolist()) <= {0.0, 2.0}
    assert int(((t[:, 0] > 0) != (i[:, 0] > 0)).sum()) > 8                   # ... and the two segments differ
    pc2 = engine.PassCtx(ops.Seq(64, 3, 5), 2, None)
    rs2 = pc2.drop_path_rows(0.5, True, torch.device("cuda"))
    assert bool(((rs2[: 64 * 3].view(64, 3)[:, 0] > 0) == (rs2[64 * 3:].view(64, 5)[:, 0] > 0)).all())  # joint pass: shared


@pytest.mark.gpu
def test_planned_droppath_sites(mods):
    """PassCtx.plan_drop_path: all sites of a pass from ONE launch; same contract as the per-site kernel (one draw per
    sample [and segment], values in {0, 1/keep}, a zero-probability site returns None but keeps its slot)."""
    engine = importlib.import_module("vl_merging_amd.engine")
    ops = importlib.import_module("vl_merging_amd.ops")
    for indep in (False, True):
        pc = engine.PassCtx(ops.Seq(64, 3, 5), 2, None)
        pc.independent_segments = indep
        pc.plan_drop_path([0.0, 0.0, 0.5, 0.5, 0.2, 0.2])
        dev = torch.device("cuda")
        assert pc.drop_path_rows(0.0, True, dev) is None and pc.drop_path_rows(0.0, True, dev) is None
        a, b = pc.drop_path_rows(0.5, True, dev), pc.drop_path_rows(0.5, True, dev)
        c = pc.drop_path_rows(0.2, True, dev)
        assert a.data_ptr() != b.data_ptr() and pc._dp_all.shape == (6, 64 * 8)
        for r, keep in ((a, 0.5), (b, 0.5), (c, 0.8)):
            t, i = r[: 64 * 3].view(64, 3), r[64 * 3:].view(64, 5)
            assert bool((t == t[:, :1]).all()) and bool((i == i[:, :1]).all())
            vals = set(round(v, 5) for v in r.unique().tolist())
            assert vals <= {0.0, round(1.0 / keep, 5)}
            same = bool(((t[:, 0] > 0) == (i[:, 0] > 0)).all())
            assert same != indep or (indep and not same)
        assert not torch.equal(a, b)  # different sites, different draws
        assert 0.6 < float((c > 0).float().mean()) < 0.95
        # a probability that does not match the plan falls back to the single-site kernel
        d = pc.drop_path_rows(0.3, True, dev)
        assert d.shape == (64 * 8,)


# ------------------------------------------------------------------------------------------------------------------
# The benchmarked width (hidden 768, 12 heads, 384^2: N = 617 tokens, R = 2294 table rows) against the reference, with
# the shipped engine options ON (fused 4B-sample joint pass, unimodal pair pass, dense fp16 bias, transposed shadows).
#
# Tolerances, next to what the reference's OWN reduced-precision path (fp16 autocast, run.py precision=16; run on the
# CPU by make_golden.py model_base -> tests/golden/amp_reference_errors.json) deviates from its fp32 path on the same
# quantities at this width:
#   quantity                                 fp16-autocast reference    this engine's bound
#   features (max err / max |ref|)           <= 1.7e-3                  1.7e-2 (1.5 x the measured 1.14e-2)
#   losses (absolute)                        <= 6e-4                    3e-2 (total 5e-2)
#   logits (max err / max |ref|)             <= 1e-3                    2e-2   (1.5 x the measured 1.36e-2, itm)
#   gradient norm per tensor, norm > 0.05    <= 0.5 %  (median 5e-5)    2.5 %  (1.5 x the measured 1.5 %, median 0.1 %)
#   gradient norm per tensor, norm <= 0.05   up to 6 % (ufo) / 114 % (all_moe: the near-zero contrastive gradients)   20 %
# fp16 keeps 11 significant bits, the engine's bf16 activations 8 (8x coarser per rounding), and the engine rounds
# between every pair of kernels where autocast only rounds GEMM operands; what the engine actually reaches is written
# to gpurun_out/parity_errors.json by every run of this file (MEASURED below) and quoted in DESIGN.md.
MEASURED = {}


@pytest.fixture(scope="module", autouse=True)
def _dump_measured():
    yield
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(os.path.join("gpurun_out", "parity_errors.json"), "w") as f:
            json.dump(MEASURED, f, indent=1, sort_keys=True)
    except OSError:
        pass


def build_base(mods, arch, golden_dir, losses, tag=None, max_vl=40, train=False, **over):
    cfgmod, vm = mods
    cfg = cfgmod.make_config(arch, vit="vit_base_patch16_384", hidden_size=768, num_heads=12, vocab_size=1024,
                             max_text_len=40, patch_size=16, vlffn_start_layer_index=10, image_size=384,
                             max_vl_text_len=max_vl, tasks=["vl"] if max_vl else None,
                             loss_names=cfgmod._loss_names(losses), **over)
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    if tag is not None:
        meta = json.load(open(os.path.join(golden_dir, f"keys_{tag}.json")))
        sd = {k: torch.from_numpy(det_array(k, s)) for k, (s, dt) in meta.items()
              if dt.startswith("float") and "index" not in k and "mask_for" not in k and not k.startswith(("train_", "val_"))}
        res = model.load_state_dict(sd, strict=False)
        assert not res.unexpected_keys, res.unexpected_keys
    model = model.cuda()
    model.train(train)
    model.setup_engine()
    return model


def check_grad_summary(model, gs):
    named = dict(model.named_parameters())
    bad, rels, floored = [], [], []
    for n, v in gs.items():
        g = named[n].grad
        if v is None:
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        nrm = float(g.double().norm())
        rels.append((abs(nrm - v[0]) / (v[0] + 1e-12), v[0]))
        cls = grad_norm_class(nrm, v[0])
        if cls is None:
            bad.append((n, nrm, v[0]))
        elif cls == "floor":
            floored.append({"tensor": n, "got": nrm, "reference": v[0]})
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0]
    big = [r for r, n0 in rels if n0 > 0.05]
    small = [r for r, n0 in rels if n0 <= 0.05]
    if floored:  # never silently: the tensors whose gradient norm passes ONLY through the absolute floor, by name
        print("gradient norms accepted through the %.1e absolute floor only:" % FLOOR)
        for f in floored:
            print("   %-60s got %.3e  reference %.3e" % (f["tensor"], f["got"], f["reference"]))
    # the floor is for near-zero SCALAR-like gradients (logit scales at B = 2): a weight matrix down there is a bug
    assert all(f["reference"] <= 4 * FLOOR for f in floored), floored
    MEASURED.setdefault(test, {}).update({"grad_norm_floor_only": floored})
    MEASURED.setdefault(test, {}).update({"grad_norm_rel_max(norm>0.05)": max(big) if big else 0.0,
                                          "grad_norm_rel_median(norm>0.05)": float(np.median(big)) if big else 0.0,
                                          "grad_norm_rel_max(norm<=0.05)": max(small) if small else 0.0})
    assert not bad, bad[:10]


@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_base_width_matches_reference_golden(mods, golden_dir, arch):
    gold = np.load(os.path.join(golden_dir, f"model_base_{arch}.npz"))
    model = build_base(mods, arch, golden_dir, {"itm": 1, "mlm": 1, "ifm": 1}, tag=f"base_{arch}")
    assert model.fuse_joint_passes and model.fuse_unimodal_passes  # the shipped defaults are what is checked
    engine = importlib.import_module("vl_merging_amd.engine")
    assert engine._DENSE_BIAS
    batch = gpu_batch(det_batch(2, 384, 40, 1024, seed=4321))
    step = int(gold["img_rows"])
    with torch.no_grad():
        r = model.infer(batch, mask_text=False)
        feat_close(r["text_feats"], gold["infer/text_feats"], "text_feats")
        feat_close(r["image_feats"][:, ::step], gold["infer/image_feats"], "image_feats")
        feat_close(r["cls_feats"], gold["infer/cls_feats"], "cls_feats")
        pi, pt = model.infer_unimodal_pair(batch, with_vlffn=True)   # what compute_ifm runs by default
        feat_close(pi["image_feats"][:, ::step], gold["infer_image/image_feats"], "pair image_feats")
        feat_close(pt["text_feats"], gold["infer_text/text_feats"], "pair text_feats")
        for k in ("cls_feats", "cls_vlffn_feats"):
            feat_close(pi[k], gold["infer_image/" + k], "pair image " + k)
            feat_close(pt[k], gold["infer_text/" + k], "pair text " + k)
        ri, rt = model.infer_image(batch), model.infer_text(batch)   # the reference-shaped separate passes
        feat_close(ri["image_feats"][:, ::step], gold["infer_image/image_feats"], "image_feats (separate)")
        feat_close(rt["text_feats"], gold["infer_text/text_feats"], "text_feats (separate)")
    model.zero_grad()
    mods[1].vilt_utils.set_task(model)
    ret = model({"vl": batch})       # ifm through the pair pass, mlm + itm through ONE 4B-sample pass
    total = sum(v for k, v in ret.items() if "loss" in k)
    total.backward()
    torch.cuda.synchronize()
    for k in ("mlm_loss", "ifm_loss", "itm_loss"):
        assert abs(float(ret[k]) - float(gold["step/" + k])) <= 3e-2, (k, float(ret[k]), float(gold["step/" + k]))
    assert abs(float(total) - float(gold["step/total_loss"])) <= 5e-2
    feat_close(ret["mlm_logits"][..., ::int(gold["mlm_cols"])], gold["step/mlm_logits"], "mlm logits", tol=2e-2)
    feat_close(ret["itm_logits"], gold["step/itm_logits"], "itm logits", tol=2e-2)
    feat_close(ret["ifm_i2t_logits"], gold["step/ifm_i2t_logits"], "ifm logits", tol=2e-2)
    check_grad_summary(model, json.loads(str(gold["step/grad_summary"])))
    named = dict(model.named_parameters())
    for key in gold.files:
        if key.startswith("step/grad/"):
            n = key[len("step/grad/"):]
            ref = torch.from_numpy(gold[key])
            got = named[n].grad.float().cpu()
            if n == "relative_position_bias_table":
                got = got[::8]
            grad_elementwise_close(got, ref, n)


def test_tolerances_are_stated_next_to_amp(golden_dir):
    """The table above quotes tests/golden/amp_reference_errors.json: keep the two in step."""
    amp = json.load(open(os.path.join(golden_dir, "amp_reference_errors.json")))
    for arch, a in amp.items():
        assert max(v for k, v in a.items() if k.startswith("feat/")) <= 1.7e-3
        assert max(v for k, v in a.items() if k.startswith("loss/")) <= 6e-4
        assert max(v for k, v in a.items() if k.startswith("logits/")) <= 1e-3
        assert a["grad_norm_rel/max_norm_gt_0.05"] <= 5e-3 and a["grad_norm_rel/median_norm_gt_0.05"] <= 1e-4
    assert amp["ufo"]["grad_norm_rel/max_norm_le_0.05"] <= 0.07 and amp["all_moe"]["grad_norm_rel/max_norm_le_0.05"] <= 1.2


def test_irtr_on_merged_weights_base_width(mods, golden_dir):
    """configs[4] at base size: all_moe weights -> merge_weights on the HIP kernel (bit-exact vs the reference's merged
    tensors) -> ufo model, irtr objective at 384^2 (B = 3): loss, logits, features, gradient norms vs the reference."""
    cfgmod, vm = mods
    gold = np.load(os.path.join(golden_dir, "irtr_merged_base.npz"))
    meta = json.load(open(os.path.join(golden_dir, "keys_base_irtr_all_moe.json")))
    import hashlib
    sd = {k: torch.from_numpy(det_array(k, s)).cuda() for k, (s, dt) in meta.items()
          if dt.startswith("float") and "index" not in k and "mask_for" not in k and not k.startswith(("train_", "val_"))}
    model = build_base(mods, "ufo", golden_dir, {"irtr": 1, "itm": 0, "mlm": 0, "ifm": 0}, max_vl=None, merge_ratio=0.5)
    merged = model.merge_weights(sd)
    torch.cuda.synchronize()
    for k in gold.files:
        if k.startswith("merged_sha/"):
            n = k[len("merged_sha/"):]
            assert hashlib.sha256(merged[n].cpu().numpy().tobytes()).hexdigest() == str(gold[k]), n
    res = model.load_state_dict(merged, strict=False)
    assert not [m for m in res.missing_keys if "index" not in m and "position_ids" not in m and "mask_for" not in m], res.missing_keys
    model._ensure_engine()
    batch = gpu_batch(det_batch(3, 384, 40, 1024, seed=99))
    model.zero_grad()
    mods[1].vilt_utils.set_task(model)
    ret = model(batch)
    ret["irtr_loss"].backward()
    torch.cuda.synchronize()
    assert abs(float(ret["irtr_loss"]) - float(gold["irtr_loss"])) <= 2e-2
    feat_close(ret["irtr_i2t_logits"], gold["irtr_i2t_logits"], "irtr logits", tol=2e-2)
    with torch.no_grad():
        feat_close(model.infer_image_ft(batch)["cls_feats"], gold["img_cls_feats"], "img cls")
        feat_close(model.infer_text_ft(batch)["cls_feats"], gold["txt_cls_feats"], "txt cls")
    check_grad_summary(model, json.loads(str(gold["grad_summary"])))


def _oracle_state(model):
    return {k: v.detach().float().cpu().clone() for k, v in model.state_dict().items() if v.is_floating_point()}


def _oracle_index(model):
    return {k: getattr(model, k).cpu() for k in ("relative_position_index", "text_relative_position_index",
                                                 "text_imag_relative_position_index")}


def test_full_size_forward_is_reproducible_at_two_workgroups_per_cu(mods, golden_dir):
    """B = 16 at 384^2: 480 forward-attention workgroups, two per CU.  The hand-placed forward once returned different rows of
    the odd samples from run to run at exactly this occupancy (a buffer load whose lanes are all out of range retires out of
    order and let a counted s_waitcnt pass early: docs/experiments.md, round 6) while every smaller case stayed bit-exact.
    Three eval passes of one model on one batch must agree bit for bit."""
    torch.manual_seed(11)
    model = build_base(mods, "ufo", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1}, tag=None, max_vl=40, train=False)
    batch = gpu_batch(det_batch(16, 384, 40, 1024, seed=77))
    outs = []
    with torch.no_grad():
        for _ in range(3):
            got = model.infer(batch)
            outs.append((got["text_feats"].clone(), got["image_feats"].clone()))
    torch.cuda.synchronize()
    for t, i in outs[1:]:
        assert torch.equal(t, outs[0][0]) and torch.equal(i, outs[0][1])
    assert bool(torch.isfinite(outs[0][0].float()).all()) and bool(torch.isfinite(outs[0][1].float()).all())


@pytest.mark.parametrize("arch,B,losses", [("all_moe", 22, {"itm": 1, "mlm": 1, "ifm": 1}),
                                           ("ufo", 22, {"itm": 1, "mlm": 1, "ifm": 1}),  # configs[1]: the bench line's workload
                                           ("ufo", 20, {"irtr": 1, "itm": 0, "mlm": 0, "ifm": 0})])
def test_full_size_step_properties(mods, golden_dir, arch, B, losses):
    """configs[2] (all_moe 384^2, B = 22, mlm + itm + ifm), configs[1] (ufo, same tasks: the bench line's workload) and
    configs[4] (irtr 384^2, B = 20) at FULL size, train mode,
    two optimizer steps.  Size-independent properties: (1) batch independence - rows of the full-batch eval pass equal
    the CPU oracle run on two of its samples alone (fp tolerance 3e-2 of the feature scale); (2) losses finite and,
    at random init on a fixed batch, lower after two AdamW steps; (3) every parameter the step's passes use receives a
    finite non-zero gradient, unused experts receive none and stay bit-identical (HF AdamW skips grad-less params)."""
    from oracle import vlmo_ref as R
    irtr = "irtr" in losses and losses["irtr"]
    model = build_base(mods, arch, golden_dir, losses, tag=None, max_vl=None if irtr else 40, train=False)
    torch.manual_seed(3)
    nb = det_batch(B, 384, 40, 1024, seed=2024 + B)
    batch = gpu_batch(nb)
    # (1) batch independence against the oracle on samples {0, B-1}
    pick = [0, B - 1]
    osd, oidx = _oracle_state(model), _oracle_index(model)
    with torch.no_grad():
        sub = {k: torch.from_numpy(v[pick]) for k, v in nb.items()}
        if irtr:
            got_i, got_t = model.infer_unimodal_pair(batch, with_vlffn=False)
            want_i = R.infer_image(osd, R.Arch(arch), oidx, sub["image"], vlffn=False)["cls_feats"]
            want_t = R.infer_text(osd, R.Arch(arch), oidx, sub["text_ids"], sub["text_masks"], vlffn=False)["cls_feats"]
            feat_close(got_i["cls_feats"][pick], want_i, "image cls of samples 0 / B-1")
            feat_close(got_t["cls_feats"][pick], want_t, "text cls of samples 0 / B-1")
        else:
            got = model.infer(batch)
            want = R.infer(osd, R.Arch(arch), oidx, sub["text_ids"], sub["text_masks"], sub["image"])
            feat_close(got["text_feats"][pick], want["text_feats"], "text_feats of samples 0 / B-1")
            feat_close(got["image_feats"][pick], want["image_feats"], "image_feats of samples 0 / B-1")
    # (2) + (3) three train-mode steps (constant lr: the configs' warm-up starts at lr = 0), judged by the deterministic
    # eval-mode loss of the same batch before and after (mlm for pre-training: no sampled negatives in it)
    probe = "irtr_loss" if irtr else "mlm_loss"

    def eval_loss():
        model.eval()
        mods[1].vilt_utils.set_task(model)
        with torch.no_grad():
            out = model(batch if irtr else {"vl": batch})
        model.train()
        return float(out[probe])

    loss_before = eval_loss()
    model.train()
    model.hparams.config["warmup_steps"] = 0
    (opt,), (sch,) = mods[1].vilt_utils.set_schedule(model, max_steps=100)
    before = {n: p.detach().clone() for n, p in model.named_parameters()}
    losses_seen = []
    for it in range(3):
        loss = model.training_step(batch if irtr else {"vl": batch})
        loss.backward()
        if it == 0:
            torch.cuda.synchronize()
            gn = {n: float(p.grad.double().norm()) for n, p in model.named_parameters()}
        opt.step()
        sch["scheduler"].step()
        losses_seen.append(float(loss))
    torch.cuda.synchronize()
    assert all(np.isfinite(losses_seen)), losses_seen
    loss_after = eval_loss()
    assert loss_after < loss_before, (probe, loss_before, loss_after, losses_seen)
    assert all(np.isfinite(v) for v in gn.values())
    used = [n for n, v in gn.items() if v > 0]
    unused = [n for n, v in gn.items() if v == 0]
    if irtr:   # text-only + image-only passes of a ufo model: every block tensor is used, the joint-pass heads are not
        assert all(n in used for n in gn if n.startswith("transformer.blocks."))
        assert "pooler.dense.weight" in unused
    elif arch == "ufo":  # configs[1]: one shared expert per layer serves every pass; the heads of all three tasks are used
        assert all(n in used for n in gn if n.startswith("transformer.blocks."))
        assert "mlm_score.decoder.weight" in used and "itm_score.fc.weight" in used and "ifm_text_proj.fc.weight" in used
        assert "transformer.mask_token" in unused
    else:      # all_moe pre-training: v / l experts in every layer, vl experts from layer 10 on
        assert "transformer.blocks.3.mlp.v.fc1.weight" in used and "transformer.blocks.3.mlp.l.fc1.weight" in used
        assert "transformer.blocks.11.mlp.vl.fc2.weight" in used
        assert "transformer.mask_token" in unused
    for n in unused:
        assert torch.equal(dict(model.named_parameters())[n].detach(), before[n]), n + " moved without a gradient"
    moved = [n for n in used if not torch.equal(dict(model.named_parameters())[n].detach(), before[n])]
    assert len(moved) >= 0.95 * len(used)


def test_zero_gradient_is_not_no_gradient(mods, golden_dir):
    """HF AdamW skips a parameter only when `p.grad is None` (vilt_utils.py:314-317).  A DropPath draw that drops every
    sample's branch gives the branch's tensors an EXACT-ZERO gradient, which the reference still treats as a gradient:
    Adam's update is 0 but the decoupled weight decay applies.  Here: every DropPath site with p > 0 drops the whole
    batch in step 1 -- the optimizer's inactive set must be the structural one (what no pass reaches) and a dropped
    block's decayed weights must have moved by exactly the decay."""
    model = build(mods, "all_moe", "tiny_all_moe", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1}, train=True)
    model.droppath_uniform_source = lambda pc, S, streams: torch.ones(streams, S, pc.seq.B)  # u >= keep: dropped
    model.hparams.config["warmup_steps"] = 0
    (opt,), (sch,) = mods[1].vilt_utils.set_schedule(model, max_steps=100)
    batch = gpu_batch(det_batch(4, 224, 40, 1024, seed=5))
    named = dict(model.named_parameters())
    w = named["transformer.blocks.5.mlp.v.fc1.weight"]
    b = named["transformer.blocks.5.mlp.v.fc1.bias"]
    w0, b0 = w.detach().clone(), b.detach().clone()
    loss = model.training_step({"vl": batch})
    loss.backward()
    torch.cuda.synchronize()
    assert float(w.grad.abs().max()) == 0.0, "the injected draw must zero block 5's gradients"
    assert float(named["transformer.blocks.0.mlp.v.fc1.weight"].grad.abs().max()) > 0  # block 0: DropPath p = 0
    opt.step()
    torch.cuda.synchronize()
    inactive = set(opt.inactive_parameters())
    assert "transformer.mask_token" in inactive and "text_embeddings.position_embeddings.weight" in inactive
    assert not [n for n in inactive if n.startswith("transformer.blocks.")], sorted(inactive)
    g = [g for g in opt.param_groups if any(lo <= model._flat.offsets["transformer.blocks.5.mlp.v.fc1.weight"][0] < hi
                                            for lo, hi in g["ranges"])]
    assert len(g) == 1 and g[0]["weight_decay"] > 0
    want = w0 - g[0]["lr"] * g[0]["weight_decay"] * w0   # Adam update 0 / (0 + eps) = 0, then p -= lr * wd * p
    assert torch.allclose(w.detach(), want, rtol=0, atol=1e-9) and not torch.equal(w.detach(), w0)
    assert torch.equal(b.detach(), b0)                   # bias: no decay group, zero update


@pytest.mark.parametrize("front", ["fused", "stock"])
@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_train_mode_step_with_injected_masks(mods, golden_dir, arch, front):
    """TRAIN mode (DropPath + text-embedding dropout live) against the REFERENCE's train-mode step on the same injected
    masks (tests/golden/train_tiny_*.npz, oracle/detweights.py::det_keep / det_dropout_mask).  The engine runs its
    shipped pass structure (pair pass + fused 4B pass), so the masks are re-keyed from the reference's
    (pass tag, site, sample) to the engine's (pass, site, stream, sample).  front = "fused": the text rows come from the one-launch
    front end (csrc/frontops.hip) with the keep mask injected through `dropout_source`; "stock": the torch module path with a
    replaced dropout module (what a caller with a custom dropout gets)."""
    from oracle.detweights import det_keep, det_dropout_mask
    gold = np.load(os.path.join(golden_dir, f"train_tiny_{arch}.npz"))
    model = build(mods, arch, f"tiny_{arch}", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1}, train=True)
    B, T, D = 2, 40, 192

    def tags_of(pc):
        if pc.independent_segments:
            return [["txt"] * pc.seq.B, ["img"] * pc.seq.B]           # stream 0 = text rows, stream 1 = image rows
        assert pc.seq.B == 4 * B
        return [[t for t in ("mlm", "pos", "negimg", "negtxt") for _ in range(B)]]

    def uniforms(pc, S, streams):
        tags = tags_of(pc)
        assert len(tags) == streams
        u = torch.ones(streams, S, pc.seq.B)
        ref_site = 0
        for s, prob in enumerate(pc._dp_sites):
            if prob <= 0.0:
                continue            # nn.Identity in the reference: no draw, no site number
            for st in range(streams):
                for b in range(pc.seq.B):
                    k = det_keep(tags[st][b], ref_site, B, 1.0 - prob)[b % B]
                    u[st, s, b] = 0.0 if k > 0 else 1.0
            ref_site += 1
        return u

    model.droppath_uniform_source = uniforms
    p = model.text_embeddings.dropout.p
    calls = []

    class DetDropout(torch.nn.Module):
        def forward(self, x):
            n = x.shape[0]
            tags = ["txt"] * n if n == B else [t for t in ("mlm", "pos", "negimg", "negtxt") for _ in range(B)]
            calls.append(n)
            m = torch.stack([torch.from_numpy(det_dropout_mask(tags[b], b % B, T, D, 1.0 - p)) for b in range(n)])
            return x * m.to(x.device) / (1.0 - p)

    def keep_mask(n, T_, D_):
        tags = ["txt"] * n if n == B else [t for t in ("mlm", "pos", "negimg", "negtxt") for _ in range(B)]
        calls.append(n)
        return torch.stack([torch.from_numpy(det_dropout_mask(tags[b], b % B, T_, D_, 1.0 - p)) for b in range(n)])

    if front == "fused":
        model.text_embeddings.dropout_source = keep_mask
        assert model._text_spec(torch.zeros(B, T, dtype=torch.int64, device="cuda")) is not None
        calls.clear()
    else:
        model.text_embeddings.dropout = DetDropout()
    batch = gpu_batch(det_batch(2, 224, 40, 1024, seed=1234))
    model.zero_grad()
    mods[1].vilt_utils.set_task(model)
    ret = model({"vl": batch})
    total = sum(v for k, v in ret.items() if "loss" in k)
    total.backward()
    torch.cuda.synchronize()
    assert calls == [B, 4 * B], calls
    for k in ("mlm_loss", "ifm_loss", "itm_loss"):
        assert abs(float(ret[k]) - float(gold["step/" + k])) <= 3e-2, (k, float(ret[k]), float(gold["step/" + k]))
    assert abs(float(total) - float(gold["step/total_loss"])) <= 5e-2
    eval_gold = np.load(os.path.join(golden_dir, f"model_tiny_{arch}.npz"))
    assert abs(float(gold["step/total_loss"]) - float(eval_gold["step/total_loss"])) > 1e-3
    feat_close(ret["mlm_logits"][..., ::int(gold["mlm_cols"])], gold["step/mlm_logits"], "mlm logits", tol=2e-2)
    feat_close(ret["itm_logits"], gold["step/itm_logits"], "itm logits", tol=2e-2)
    check_grad_summary(model, json.loads(str(gold["step/grad_summary"])))


@pytest.mark.gpu
def test_infer_with_precomputed_image_embeds(mods, golden_dir):
    """infer(image_embeds=, image_masks=) (reference vilt_module.py:1092-1108: visual_embed's output handed in): the same
    features as the pass that embeds the image itself; a dropped image token (mask 0) is a key nobody attends to -- the
    result equals the reference restatement of that mask in the torch oracle of the attention test, here checked through
    the property that the text features then differ and the kept rows stay finite.  (The reference's own path ends in an
    UnboundLocalError at its result dict, `"image": img`; here that entry is None.)"""
    model = build(mods, "ufo", "tiny_ufo", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1})
    batch = gpu_batch(det_batch(3, 224, 40, 1024, seed=17))
    with torch.no_grad():
        want = model.infer(batch)
        emb, masks, _, _ = model.transformer.visual_embed(batch["image"][0], max_image_len=model.hparams.config["max_image_len"])
        got = model.infer(batch, image_embeds=emb, image_masks=masks)
        for k in ("text_feats", "image_feats", "cls_feats", "raw_cls_feats"):
            # not bit-equal: the pass that embeds the image adds (conv bias + token type) in the patch-embed GEMM's epilogue
            # (engine.pass_rows), visual_embed's output gets the token type added afterwards -- one fp32 rounding apart
            feat_close(got[k], want[k].float().cpu(), "precomputed embeds " + k, tol=5e-3)
        assert got["image"] is None and got["image_labels"] is None and got["patch_index"] is None
        m2 = masks.clone()
        m2[:, -20:] = 0
        cut = model.infer(batch, image_embeds=emb, image_masks=m2)
        assert torch.isfinite(cut["text_feats"].float()).all()
        assert float((cut["text_feats"].float() - want["text_feats"].float()).abs().max()) > 1e-3
        # the dropped keys really are invisible: their embeddings may be anything
        emb2 = emb.clone()
        emb2[:, -20:] = 123.0
        cut2 = model.infer(batch, image_embeds=emb2, image_masks=m2)
        assert torch.equal(cut2["text_feats"], cut["text_feats"])
        assert torch.equal(cut2["image_feats"][:, :-20], cut["image_feats"][:, :-20])
    with pytest.raises(ValueError):
        model.infer(batch, image_embeds=emb)


def test_dense_bias_is_cached_on_the_table_version(mods, golden_dir):
    """vlm_bias_dense output (the tiled fp16 relative-position bias of all layers and heads) is rebuilt only when the table
    changed: a no_grad sweep over many batches (compute_irtr_recall) builds it once per pass geometry; an optimizer step, a
    reload and an in-place edit of the table each invalidate it -- and the outputs follow the new table."""
    eng = importlib.import_module("vl_merging_amd.engine")
    model = build(mods, "ufo", "tiny_ufo", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1})
    batch = gpu_batch(det_batch(2, 224, 40, 1024, seed=77))
    st = eng._DENSE_STATS

    def sweep(n):
        b0, h0 = st["built"], st["hits"]
        with torch.no_grad():
            outs = [model.infer(batch, mask_text=False)["cls_feats"].clone() for _ in range(n)]
        return st["built"] - b0, st["hits"] - h0, outs

    sweep(1)                                   # whatever state earlier tests left: make the cache current
    built, hits, o1 = sweep(3)
    assert built == 0 and hits > 0 and hits % 3 == 0  # three passes, no rebuild
    k = hits // 3                              # dense tables per pass (one per attention mode the layers use)
    assert torch.equal(o1[0], o1[2])
    # (1) an in-place torch edit of the table
    with torch.no_grad():
        model.relative_position_bias_table.mul_(1.5)
    built, hits, o2 = sweep(2)
    assert built == k and hits == k
    assert not torch.equal(o2[0], o1[0])
    # (2) an optimizer step (the HIP kernel writes the table through a raw pointer: torch's version counter does not move)
    model.train()
    model.hparams.config["warmup_steps"] = 0
    (opt,), _ = mods[1].vilt_utils.set_schedule(model, max_steps=10)
    v0 = model.relative_position_bias_table._version
    model.training_step({"vl": batch}).backward()
    opt.step()
    model.eval()
    built, hits, o3 = sweep(2)
    assert built == k and hits == k, (built, hits, v0, model.relative_position_bias_table._version)
    assert not torch.equal(o3[0], o2[0])
    # (3) a reload
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    sd["relative_position_bias_table"] = sd["relative_position_bias_table"] * 0.5
    model.load_state_dict(sd)
    built, hits, o4 = sweep(2)
    assert built == k and hits == k
    assert not torch.equal(o4[0], o3[0])


def test_dense_bias_cache_dies_with_its_model(mods, golden_dir):
    """The cross-pass cache hangs off the model's FlatParams: a model freed and rebuilt in the same process (sequential
    evaluations of merged checkpoints) starts with an empty cache whatever ids and device pointers the allocator recycles,
    and a temporary index (name None in get_rel_pos_bias) never enters it."""
    import gc
    eng = importlib.import_module("vl_merging_amd.engine")
    batch = gpu_batch(det_batch(2, 224, 40, 1024, seed=78))
    st = eng._DENSE_STATS
    feats, serials = [], []
    for scale in (1.0, 2.0):
        model = build(mods, "ufo", "tiny_ufo", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1})
        with torch.no_grad():
            model.relative_position_bias_table.mul_(scale)
        flat = model.relative_position_bias_table._vlm_flat
        assert flat.dense_cache == {}
        serials.append(flat.serial)
        b0 = st["built"]
        with torch.no_grad():
            feats.append(model.infer(batch, mask_text=False)["cls_feats"].clone())
        assert st["built"] > b0 and len(flat.dense_cache) > 0
        # a temporary index: same values as the model's buffer, a fresh tensor -> never cached
        tmp = model.text_imag_relative_position_index.clone()
        n0 = len(flat.dense_cache)
        rp = model.get_rel_pos_bias(tmp)
        assert rp.cache_tag is None and rp.cache is None and len(flat.dense_cache) == n0
        del model, flat, rp
        gc.collect()
        torch.cuda.empty_cache()
    assert serials[0] != serials[1]
    assert not torch.equal(feats[0], feats[1])


def test_ten_step_trajectory_follows_the_oracle(mods, golden_dir):
    """TEN optimizer steps (eval-mode forward: no RNG; B = 2: the hard negatives are forced) on the GPU engine -- HIP kernels,
    bf16 operands, FusedAdamW -- and in the oracle: fp32 autograd through oracle/vlmo_ref.pretrain_step on the CPU with
    oracle/adamw_hf4.AdamWHF4 (float64 restatement of the published transformers-4.x rule; UNPINNED, see its header) under the
    parameter groups and schedule that tests/golden/schedule_groups.json pins to the reference's own set_schedule.
    Checked: the loss of every step (3e-2, the single-step bound) and, per tensor, how far the ten-step weight change
    dW = W_10 - W_0 is from the oracle's.  Adam divides each gradient element by its own running magnitude: every element moves by
    about lr per step in the direction of its gradient's SIGN, however small the gradient.  An element whose gradient is inside
    the bf16 noise (the single-step element-wise error measured here is 4-10 % of the tensor's largest entry) therefore takes
    steps of arbitrary sign on both sides, and over ALL elements the ten-step drift is 15-30 % for most tensors (recorded in
    parity_errors.json; the reference's own fp16-autocast run would show the same).  The drift is therefore asserted on the
    elements whose gradient is resolved -- the oracle's mean |g| over the ten steps at or above a quarter of its largest entry in
    the tensor -- and must stay within 2 % there (measured <= 0.7 %; 25 % for the two scalar logit scales, whose B = 2 gradient
    is below every reduced-precision path's noise floor: measured 12.5 %)."""
    import copy
    from oracle import vlmo_ref as R
    from oracle.adamw_hf4 import AdamWHF4, polynomial_decay_with_warmup
    cfgmod, vm = mods
    vu = vm.vilt_utils
    model = build(mods, "ufo", "tiny_ufo", golden_dir, {"itm": 1, "mlm": 1, "ifm": 1})
    cfg = model.hparams.config
    STEPS, MAX_STEPS = 10, 100
    (opt,), (sch,) = vu.set_schedule(model, max_steps=MAX_STEPS)
    nb = det_batch(2, 224, 40, 1024, seed=1234)
    batch = gpu_batch(nb)
    w0 = {n: p.detach().cpu().clone() for n, p in model.named_parameters()}
    # ---- oracle side --------------------------------------------------------------------------------------------------------
    idx = {k: getattr(model, k).cpu() for k in ("relative_position_index", "text_relative_position_index",
                                                "text_imag_relative_position_index")}
    heads = vu.head_names(cfg)
    spec = vu.group_spec(cfg)
    groups = {n: (spec[vu.param_group_of(n, heads)][1], spec[vu.param_group_of(n, heads)][0]) for n in w0}
    warm = cfg["warmup_steps"]
    warm = int(MAX_STEPS * warm) if isinstance(warm, float) else warm
    master = {n: w.double().numpy().copy() for n, w in w0.items()}
    oopt = AdamWHF4(master, groups, betas=(0.9, cfg["beta_2"]), eps=1e-8)
    ob = {k: torch.from_numpy(v) for k, v in nb.items()}
    arch = R.Arch("ufo", hidden=192, heads=3)
    extra = {k: v.detach().cpu() for k, v in model.state_dict().items() if k not in w0 and v.is_floating_point()}
    gpu_losses, ora_losses, gabs = [], [], {}
    torch.set_num_threads(min(16, torch.get_num_threads()))
    for it in range(STEPS):
        loss = model.training_step({"vl": batch}, it)
        loss.backward()
        opt.step()
        sch["scheduler"].step()
        gpu_losses.append(float(loss.detach()))
        osd = {n: torch.from_numpy(master[n]).float().requires_grad_(True) for n in master}
        full = dict(extra)
        full.update(osd)
        ref = R.pretrain_step(full, arch, idx, ob)
        ref["total_loss"].backward()
        ora_losses.append(float(ref["total_loss"].detach()))
        fac = polynomial_decay_with_warmup(it, warm, MAX_STEPS, cfg["learning_rate"], cfg["end_lr"], cfg["decay_power"])
        og = {n: (t.grad.numpy() if t.grad is not None else None) for n, t in osd.items()}
        for n, g_ in og.items():
            if g_ is not None:
                gabs[n] = gabs.get(n, 0.0) + np.abs(g_) / STEPS
        oopt.step(og, fac)
    torch.cuda.synchronize()
    test = os.environ.get("PYTEST_CURRENT_TEST", "").split("::")[-1].split(" ")[0]
    rec = MEASURED.setdefault(test, {})
    rec["loss max abs err over 10 steps"] = max(abs(a - b) for a, b in zip(gpu_losses, ora_losses))
    for a, b in zip(gpu_losses, ora_losses):
        assert abs(a - b) <= 3e-2, (gpu_losses, ora_losses)
    assert ora_losses[-1] < ora_losses[0] and gpu_losses[-1] < gpu_losses[0]
    worst_all, worst_big, n_checked = ("", 0.0), ("", 0.0), 0
    table, failed = [], []
    for n, p in model.named_parameters():
        dw_o = torch.from_numpy(master[n]).float() - w0[n]
        if float(dw_o.abs().max()) == 0.0:
            assert torch.equal(p.detach().cpu(), w0[n]), n  # no gradient on either side: untouched (no decay either)
            continue
        dw_g = p.detach().cpu() - w0[n]
        rel_all = float((dw_g - dw_o).norm() / dw_o.norm())
        ga = torch.from_numpy(np.asarray(gabs[n], dtype=np.float32))
        big = ga >= 0.25 * ga.max()
        rel_big = float((dw_g - dw_o)[big].norm() / dw_o[big].norm())
        n_checked += 1
        if rel_all > worst_all[1]:
            worst_all = (n, rel_all)
        if rel_big > worst_big[1] and p.numel() > 1:
            worst_big = (n, rel_big)
        # scalars (the two logit scales): one number whose gradient at B = 2 is below the noise floor of any reduced-precision
        # path (grad_norm_ok) -- the update's SIGN must agree, its size within 25 %
        lim = 0.25 if p.numel() == 1 else 0.02  # measured: <= 0.7 % on every tensor, 12.5 % on logit_vl_scale
        table.append((rel_big, rel_all, n))
        if rel_big > lim:
            failed.append((n, round(rel_big, 4), round(rel_all, 4), lim))
    table.sort(reverse=True)
    print("largest drifts (update-carrying elements, all elements, tensor):")
    for t in table[:12]:
        print("   %.4f  %.4f  %s" % t)
    med = sorted(t[0] for t in table)[len(table) // 2]
    rec["weight drift, elements that carry the update, median tensor"] = med
    assert not failed, failed
    rec["weight drift, elements that carry the update, worst tensor"] = worst_big[1]
    rec["weight drift, all elements, worst tensor"] = worst_all[1]
    assert n_checked > 100
    print("trajectory: loss err %.2e; drift worst (update-carrying elements) %s %.3f; (all elements) %s %.3f"
          % (rec["loss max abs err over 10 steps"], worst_big[0], worst_big[1], worst_all[0], worst_all[1]))
