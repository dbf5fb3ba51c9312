"""Pin the CPU model oracle (oracle/vlmo_ref.py) against outputs of the reference itself (tests/golden/*.npz)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vlmo_ref as R
from oracle.detweights import det_array, det_batch


def load_state(golden_dir, tag):
    meta = json.load(open(os.path.join(golden_dir, f"keys_{tag}.json")))
    sd = {}
    for k, (shape, dt) in meta.items():
        if dt.startswith("float") and "index" not in k and "mask_for" not in k:
            sd[k] = torch.from_numpy(det_array(k, shape)).requires_grad_(True)
    return sd, meta


def index_buffers(golden_dir, tag="224"):
    z = np.load(os.path.join(golden_dir, "index_buffers.npz"))
    return {k: torch.from_numpy(z[f"{k}_{tag}"]) for k in
            ("relative_position_index", "text_relative_position_index", "text_imag_relative_position_index")}


def tb(nb):
    return {k: torch.from_numpy(v) for k, v in nb.items()}


def close(a, b, tol=2e-4):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    scale = b.abs().max().item() + 1e-12
    assert err <= tol * scale + 1e-6, (err, scale)


@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_oracle_pretrain_step_matches_reference(arch, golden_dir):
    torch.manual_seed(0)
    gold = np.load(os.path.join(golden_dir, f"model_tiny_{arch}.npz"))
    sd, _ = load_state(golden_dir, f"tiny_{arch}")
    idx = index_buffers(golden_dir)
    a = R.Arch(arch, hidden=192, heads=3)
    batch = tb(det_batch(2, 224, 40, 1024, seed=1234))
    with torch.no_grad():
        r = R.infer(sd, a, idx, batch["text_ids"], batch["text_masks"], batch["image"])
        for k in ("text_feats", "image_feats", "cls_feats", "raw_cls_feats"):
            close(r[k], gold["infer/" + k])
        r = R.infer_image(sd, a, idx, batch["image"])
        for k in ("image_feats", "cls_feats", "cls_vlffn_feats"):
            close(r[k], gold["infer_image/" + k])
        r = R.infer_text(sd, a, idx, batch["text_ids"], batch["text_masks"])
        for k in ("text_feats", "cls_feats", "cls_vlffn_feats"):
            close(r[k], gold["infer_text/" + k])
        x = torch.from_numpy(det_array("probe.x", (2, 237, 192))) * 10
        mask = torch.cat([batch["text_masks"], torch.ones(2, 197, dtype=torch.long)], 1)
        bl = R.rel_pos_bias(sd, idx["text_imag_relative_position_index"], a)
        for li in (0, 11):
            close(R.block(sd, a, li, x, mask, 2, bl[li]), gold[f"block{li}/joint"])
    out = R.pretrain_step(sd, a, idx, batch)
    for k in ("mlm_loss", "ifm_loss", "itm_loss", "total_loss"):
        close(out[k].detach(), gold["step/" + k], 1e-5)
    close(out["mlm_logits"].detach(), gold["step/mlm_logits"])
    close(out["itm_logits"].detach(), gold["step/itm_logits"])
    out["total_loss"].backward()
    gs = json.loads(str(gold["step/grad_summary"]))
    for n, v in gs.items():
        if v is None:
            assert sd[n].grad is None or float(sd[n].grad.abs().max()) == 0.0, n
        else:
            g = sd[n].grad.double()
            assert abs(float(g.norm()) - v[0]) <= 2e-4 * v[0] + 1e-7, (n, float(g.norm()), v[0])
    for key in gold.files:
        if key.startswith("step/grad/"):
            close(sd[key[len("step/grad/"):]].grad, gold[key], 5e-4)


@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_oracle_irtr_and_grams_match_reference(arch, golden_dir):
    gold = np.load(os.path.join(golden_dir, f"irtr_tiny_{arch}.npz"))
    sd, _ = load_state(golden_dir, f"tiny_irtr_{arch}")
    idx = index_buffers(golden_dir)
    a = R.Arch(arch, hidden=192, heads=3)
    batch = tb(det_batch(3, 224, 40, 1024, seed=77))
    out = R.irtr_step(sd, a, idx, batch)
    close(out["irtr_loss"].detach(), gold["irtr_loss"], 1e-5)
    close(out["irtr_i2t_logits"].detach(), gold["irtr_i2t_logits"])
    out["irtr_loss"].backward()
    gs = json.loads(str(gold["grad_summary"]))
    for n, v in gs.items():
        if v is not None:
            g = sd[n].grad.double()
            assert abs(float(g.norm()) - v[0]) <= 2e-4 * v[0] + 1e-8, n
    if arch == "all_moe":
        with torch.no_grad():
            grams = R.gram_inputs(sd, a, idx, batch)
        assert sorted(grams) == json.loads(str(gold["gram_keys"]))
        summ = json.loads(str(gold["gram_summary"]))
        for k, (shape, nrm, sm) in summ.items():
            assert list(grams[k].shape) == shape
            assert abs(float(grams[k].norm()) - nrm) <= 1e-5 * nrm
        for key in gold.files:
            if key.startswith("gram/"):
                g = grams[key[5:]][:192, :192]
                close(g, gold[key], 1e-5)


def test_oracle_grams_base_width_match_reference(golden_dir):
    """The Gram hook at BASE width (hidden 768 / F 3072, 384^2, all_moe irtr model, B = 3) on the reference's own
    output (tests/golden/gram_base.npz): pins oracle/vlmo_ref.py::gram_inputs at the size configs[3] names."""
    gold = np.load(os.path.join(golden_dir, "gram_base.npz"))
    sd, _ = load_state(golden_dir, "base_irtr_all_moe")
    idx = index_buffers(golden_dir, "384")
    a = R.Arch("all_moe")
    batch = tb(det_batch(3, 384, 40, 1024, seed=99))
    with torch.no_grad():
        grams = R.gram_inputs(sd, a, idx, batch)
    assert sorted(grams) == json.loads(str(gold["gram_keys"]))
    summ = json.loads(str(gold["gram_summary"]))
    for k, (shape, nrm, sm) in summ.items():
        assert list(grams[k].shape) == shape
        assert abs(float(grams[k].norm()) - nrm) <= 1e-5 * nrm, k
    for key in gold.files:
        if key.startswith("gram/"):
            close(grams[key[5:]][:256, :256].float(), gold[key], 1e-4)


def test_index_buffers_known_answers(golden_dir):
    """sha256 / sums recorded from the reference for both resolutions."""
    import hashlib
    z = np.load(os.path.join(golden_dir, "index_buffers.npz"))
    assert int(z["relative_position_index_224"].sum()) == 14270119
    assert int(z["relative_position_index_384"].sum()) == 368828259
    assert int(z["text_relative_position_index_384"].min()) == 2370
    for k in z.files:
        if k.endswith("_sha256"):
            assert hashlib.sha256(np.ascontiguousarray(z[k[:-7]]).tobytes()).hexdigest() == str(z[k])


@pytest.mark.parametrize("arch", ["ufo", "all_moe"])
def test_oracle_train_mode_step_matches_reference(arch, golden_dir):
    """TRAIN mode: DropPath and the text-embedding dropout are live in the reference, with the injected masks of
    oracle/detweights.py (tests/golden/ref_harness.py::inject_train_masks); the oracle consumes the same masks."""
    gold = np.load(os.path.join(golden_dir, f"train_tiny_{arch}.npz"))
    sd, _ = load_state(golden_dir, f"tiny_{arch}")
    idx = index_buffers(golden_dir)
    a = R.Arch(arch, hidden=192, heads=3)
    tm = R.TrainMasks(0.1, a.L)
    assert np.allclose(tm.dpr, gold["drop_path_probs"], atol=1e-7)
    batch = tb(det_batch(2, 224, 40, 1024, seed=1234))
    out = R.pretrain_step(sd, a, idx, batch, tm=tm)
    for k in ("mlm_loss", "ifm_loss", "itm_loss", "total_loss"):
        close(out[k].detach(), gold["step/" + k], 1e-5)
    close(out["mlm_logits"].detach()[..., ::int(gold["mlm_cols"])], gold["step/mlm_logits"])
    close(out["itm_logits"].detach(), gold["step/itm_logits"])
    eval_gold = np.load(os.path.join(golden_dir, f"model_tiny_{arch}.npz"))
    assert abs(float(gold["step/total_loss"]) - float(eval_gold["step/total_loss"])) > 1e-3  # the masks did something
    out["total_loss"].backward()
    gs = json.loads(str(gold["step/grad_summary"]))
    for n, v in gs.items():
        if v is not None:
            g = sd[n].grad.double()
            assert abs(float(g.norm()) - v[0]) <= 2e-4 * v[0] + 1e-7, (n, float(g.norm()), v[0])
    for key in gold.files:
        if key.startswith("step/grad/"):
            close(sd[key[len("step/grad/"):]].grad, gold[key], 5e-4)


def test_oracle_base_width_forward_matches_reference(golden_dir):
    """The benchmarked width (hidden 768, 12 heads, 384^2: N = 617) on the reference's own outputs, forward passes of
    the ufo model (the full step at this size is checked on the GPU box, where the oracle has the cores)."""
    gold = np.load(os.path.join(golden_dir, "model_base_ufo.npz"))
    sd, _ = load_state(golden_dir, "base_ufo")
    idx = index_buffers(golden_dir, "384")
    a = R.Arch("ufo")
    batch = tb(det_batch(2, 384, 40, 1024, seed=4321))
    step = int(gold["img_rows"])
    with torch.no_grad():
        r = R.infer(sd, a, idx, batch["text_ids"], batch["text_masks"], batch["image"])
        close(r["text_feats"], gold["infer/text_feats"])
        close(r["image_feats"][:, ::step], gold["infer/image_feats"])
        close(r["cls_feats"], gold["infer/cls_feats"])
        r = R.infer_text(sd, a, idx, batch["text_ids"], batch["text_masks"])
        for k in ("text_feats", "cls_feats", "cls_vlffn_feats"):
            close(r[k], gold["infer_text/" + k])
