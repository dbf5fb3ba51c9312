"""The MI355X model (HIP kernels, bf16 GEMMs / fp32 accumulation) against the reference's golden outputs and the
fp32 oracle on the same deterministic weights and batch.

Stated tolerance (north star: "fp within a stated tol for attention/FFN"): activations are rounded to bf16 between
kernels (8 significant bits, like the reference's fp16 AMP path rounds to 11), so
   features after 12 blocks : max |err| <= 3e-2 * max|ref|
   logits / losses          : |err| <= 3e-2 (absolute, values are O(1..10))
   parameter-gradient norms : relative error <= 6e-2 per tensor (bf16 operands of the wgrad GEMMs)
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


def grad_norm_ok(got, ref):
    """6 % per tensor.  Tensors with a small gradient (norm <= 0.05: here those fed only by the B=2 contrastive
    losses, i.e. by the DIFFERENCE of two nearly identical L2-normalised features scaled by exp(logit_scale) ~ 14,
    ill-conditioned in any 8-bit-mantissa activation format, and run-to-run sensitive to the order of the fp32 atomic
    accumulations) get 20 %; near-zero scalar gradients an absolute 1e-4."""
    rel = abs(got - ref) / (ref + 1e-12)
    return rel <= 6e-2 or (ref <= 0.05 and rel <= 0.20) or abs(got - ref) <= 1e-4


def feat_close(got, ref, what, tol=3e-2):
    got = got.float().cpu()
    ref = torch.as_tensor(ref)
    err = float((got - ref).abs().max())
    scale = float(ref.abs().max())
    assert err <= tol * scale, "%s: max err %.4g vs scale %.4g" % (what, err, scale)


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
    feat_close(ret["mlm_logits"], gold["step/mlm_logits"], "mlm logits")
    feat_close(ret["itm_logits"], gold["step/itm_logits"], "itm logits", tol=5e-2)
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
            err = float((got - ref).abs().max())
            mx = float(ref.abs().max())
            floor = 1e-4 if ref.numel() == 1 else 1e-6  # near-zero scalar (logit scale) gradients: see grad_norm_ok
            assert err <= (0.2 if mx <= 0.05 else 0.15) * mx + floor, (n, err, mx)


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
    feat_close(ret["irtr_i2t_logits"], gold["irtr_i2t_logits"], "irtr logits")
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
    assert torch.equal(model._flat.flat_b[:model._flat.numel], model._flat.flat_p[:model._flat.numel].to(torch.bfloat16))
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
    feat_close(ret[task + "_logits"], gold[task + "/logits"], task + " logits", tol=5e-2)
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
