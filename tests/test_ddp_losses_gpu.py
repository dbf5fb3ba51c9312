"""The losses that gather across ranks, with world size 2 on the REAL model (two gloo ranks on the one GPU of the test
box, fresh child processes; helper tests/helpers/ddp_gather_losses.py).  Reference: objectives.py:176-178 (hard-negative
candidates: ids, masks and raw images of every rank), :274-300 / :393-394 (contrastive features, own block first, no
autograd through the gathered copies), Lightning DDP's gradient average (run.py:263-288).

What data parallelism must reproduce, against ONE process on the concatenated 2B batch:
  * mlm / itm (means over per-rank samples): average over ranks of the gradients == the 2B gradient;
  * ifm / irtr (every rank evaluates the FULL contrastive loss but back-propagates through its own features only,
    objectives.py:277-286): average over ranks == 1/W of the 2B gradient for everything upstream of the features, and
    the full 2B gradient for the logit scales (each rank's loss sees them in full).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
HELPER = os.path.join(HERE, "helpers", "ddp_gather_losses.py")
sys.path.insert(0, os.path.join(HERE, "helpers"))


def run_pair(outdir, config):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, HELPER, outdir, config], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(outdir, "%s_rank%d.npz" % (config, r))) for r in range(2)]


_OFFSETS = []


def single_process(config):
    """(G_contrastive, G_rest, losses) of one process on the concatenated batch; flat gradients as numpy."""
    import ddp_gather_losses as H
    H.deterministic_negatives()
    model, vm = H.build_model(H.LOSSES[config], max_vl=40 if config == "pretrain" else None)
    nb = H.fixed_mask_batch(2 * H.PER)
    batch = H.gpu_rows(nb, 0, 2 * H.PER)
    f = model._flat
    grads, losses = {}, {}
    contrast = "ifm_loss" if config == "pretrain" else "irtr_loss"
    for part in ("contrastive", "rest"):
        f.flat_g.zero_()
        vm.vilt_utils.set_task(model)
        ret = model(H.wrap(config, dict(batch)))
        picked = [v for k, v in ret.items() if "loss" in k and ((k == contrast) == (part == "contrastive"))]
        if picked:
            sum(picked).backward()
        torch.cuda.synchronize()
        grads[part] = f.flat_g[:f.numel].cpu().numpy().copy()
        losses.update({k: float(v.detach()) for k, v in ret.items() if "loss" in k})
    scale_names = [n for n in f.names if n in ("logit_scale", "logit_vl_scale")]
    scale_at = [f.offsets[n][0] for n in scale_names]
    global _OFFSETS
    _OFFSETS = sorted((o, n) for n, (o, k) in f.offsets.items())
    return grads, losses, scale_at, nb


@pytest.mark.parametrize("config", ["pretrain", "irtr"])
def test_gathering_losses_two_ranks(pkg, tmp_path, config):
    orig_multinomial = torch.multinomial
    try:
        grads, losses, scale_at, nb = single_process(config)
    finally:
        torch.multinomial = orig_multinomial  # the helper's deterministic stand-in must not leak into other tests
    torch.cuda.empty_cache()
    r = run_pair(str(tmp_path), config)
    per, W = 2, 2
    # ---- (i) the gathers: own block first, the others in rank order ---------------------------------------------------
    for rank in range(W):
        order = [rank] + [x for x in range(W) if x != rank]
        rows = np.concatenate([np.arange(o * per, (o + 1) * per) for o in order])
        assert np.array_equal(r[rank]["cand_text_ids"], nb["text_ids"][rows])
        assert np.array_equal(r[rank]["cand_text_masks"], nb["text_masks"][rows])
        assert np.array_equal(r[rank]["cand_images_plain"], nb["image"][rows])  # the reference's fp32 gather
        # the prefetch ships the images as bf16: exactly the rounding the patch embedding applies anyway
        want = torch.from_numpy(nb["image"][rows]).to(torch.bfloat16).float().numpy()
        assert np.array_equal(r[rank]["cand_images"], want)
        assert np.array_equal(r[rank]["first_own"], np.repeat(np.array(order, dtype=np.float32) + 1, per)[:, None] * np.ones((1, 4), np.float32))
        assert np.array_equal(r[rank]["first_own_grad"], np.ones((per, 4), np.float32))
    # ---- (iii) losses and the averaged gradient ---------------------------------------------------------------------------
    contrast = "ifm_loss" if config == "pretrain" else "irtr_loss"
    for k, v in losses.items():
        mean = 0.5 * (float(r[0][k]) + float(r[1][k]))
        assert abs(mean - v) <= 3e-3 * max(1.0, abs(v)), (k, mean, v)
        if k == contrast:  # every rank evaluates the full contrastive loss
            assert abs(float(r[0][k]) - float(r[1][k])) <= 2e-3 * max(1.0, abs(v)), k
    want = grads["rest"] + grads["contrastive"] / W
    for at in scale_at:
        want[at] = grads["rest"][at] + grads["contrastive"][at]
    assert np.array_equal(r[0]["grad_avg"], r[1]["grad_avg"])  # both ranks hold the same all-reduced buffer
    got = r[0]["grad_avg"]
    scale = np.abs(want).max()
    err = np.abs(got - want).max()
    at = int(np.abs(got - want).argmax())
    where = [n for o, n in _OFFSETS if o <= at][-1]
    # 5 % of the largest gradient: the irtr step at random init is ill-conditioned (loss = ln 4 + 1e-3: the contrastive gradient
    # is a difference of nearly equal terms), and the worst element -- one entry of cls_token -- moves by 1-2 % of the scale in the
    # SINGLE-process run when one kernel is exchanged for an equivalent one (fused / unfused attention backward, folded / unfolded
    # LayerScale: round 5, gpurun_out/r05/call_f.txt), while both set-ups agree with the reference to the goldens' tolerance
    # The well-conditioned pretrain step keeps 2 %: a reducer / finish ordering bug (a missed finish_layerscale on one rank moves
    # the gamma gradients by about 5 %) must not hide behind the irtr case's allowance.
    assert err <= (2e-2 if config == "pretrain" else 5e-2) * scale, (err, scale, where, float(got[at]), float(want[at]))
    # the contrastive share is really there (and really scaled by 1/W): leaving it out or taking it in full must fail
    c = np.abs(grads["contrastive"]).max()
    assert c / W > 4e-2 * scale or config == "pretrain"
    if config == "irtr":
        assert np.abs(got - grads["contrastive"]).max() > 0.2 * scale
    # identical parameters on both ranks after the optimizer step
    assert np.array_equal(r[0]["params"], r[1]["params"])


def test_bf16_image_candidates_give_identical_patch_embeddings(pkg):
    """hard-negative candidate images travel as bf16 because the patch embedding rounds its input to
    bf16 anyway (im2col feeds the MFMA GEMM) -- the features of a gathered negative are BIT-identical."""
    import importlib
    engine = importlib.import_module("vl_merging_amd.engine")
    import ddp_gather_losses as H
    model, _ = H.build_model({"mlm": 1})
    g = torch.Generator(device="cuda").manual_seed(5)
    img = torch.rand(3, 3, 224, 224, device="cuda", generator=g) * 2 - 1
    pe = model.transformer.patch_embed.proj
    with torch.no_grad():
        a = engine.patch_embed(img, pe.weight, pe.bias, 16)
        b = engine.patch_embed(img.to(torch.bfloat16).float(), pe.weight, pe.bias, 16)
    assert torch.equal(a, b)
