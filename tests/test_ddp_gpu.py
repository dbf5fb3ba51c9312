"""Data-parallel equivalence on the REAL model (reference: pytorch_lightning DDP, run.py:263-288, and the `ddp_sharded`
plugin, run.py:231-232): two ranks x B samples must reproduce one process on the concatenated 2B batch.

Two gloo ranks share the one GPU of the test box (fresh child processes; RCCL itself needs one device per rank and is
exercised by the driver's multi-GPU bench).  Checked: (1) the all-reduced / reduce-scattered flat gradient, scaled by
1/world, equals the single-process gradient of the 2B batch within bf16 tolerance; (2) with all-reduce, all-reduce with the
deferred embeddings bucket, and the sharded optimizer (with and without deferral) both ranks hold BIT-IDENTICAL
parameters after three AdamW steps, and the four configurations agree with each other to what the backward's atomics
order allows; (3) the sharded optimizer holds 1/world of the Adam state.
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
HELPER = os.path.join(HERE, "helpers", "ddp_one_device.py")


def run_pair(outdir, config, extra_env=None):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, HELPER, outdir, config], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [np.load(os.path.join(outdir, "%s_rank%d.npz" % (config, r))) for r in range(2)]


def test_two_ranks_match_single_process_and_sharded_optimizer(pkg, tmp_path):
    sys.path.insert(0, os.path.join(HERE, "helpers"))
    import ddp_one_device as H
    # ---- single process, concatenated batch ---------------------------------------------------------------------
    model, vm = H.build_model()
    nb = H.fixed_mask_batch(4)
    loss = model.training_step({"vl": H.gpu_rows(nb, 0, 4)}, 0)
    loss.backward()
    torch.cuda.synchronize()
    f = model._flat
    g_ref = f.flat_g[:f.numel].cpu().numpy().copy()
    loss_ref = float(loss)
    del model
    torch.cuda.empty_cache()
    # ---- two ranks ----------------------------------------------------------------------------------------------------
    res = {c: run_pair(str(tmp_path), c) for c in ("allreduce", "allreduce_defer", "sharded", "sharded_defer")}
    scale = np.abs(g_ref).max()
    for c, (r0, r1) in res.items():
        assert abs(0.5 * (float(r0["loss0"]) + float(r1["loss0"])) - loss_ref) <= 2e-3, c
        for r in (r0, r1):
            assert bool(r["shadow_ok"]), c
            g = r["grad0"]
            if c.startswith("sharded"):
                own = r["own"]
                err = np.abs(g - g_ref)[own].max()
            else:
                err = np.abs(g - g_ref).max()
            # different batch composition -> different GEMM tiling / accumulation order of bf16 products
            assert err <= 2e-2 * scale, (c, err, scale)
        if c.startswith("sharded"):
            assert not (r0["own"] & r1["own"]).any() and (r0["own"] | r1["own"]).all()  # a partition of the buffer
            assert int(r0["state_elements"]) * 2 == int(r0["numel"]) == int(r1["numel"])
            g = np.where(r0["own"], r0["grad0"], r1["grad0"])
            assert np.abs(g - g_ref).max() <= 2e-2 * scale
    # the data-parallel invariant: after every step both ranks hold BIT-IDENTICAL parameters (same reduced gradients,
    # deterministic AdamW; sharded: every chunk comes from its one owner)
    for c, (r0, r1) in res.items():
        assert r0["params"].tobytes() == r1["params"].tobytes(), c + ": ranks diverged"
    # across configurations (separate runs) the backward's fp32 atomics order differs, so gradients agree to ~1e-6
    # relative and AdamW's normalised update (lr * m / sqrt(v), lr = 1e-4, three steps) can move an element whose
    # gradient is noise by up to 2 * lr per step; everything else agrees to rounding
    base = res["allreduce"][0]["params"]
    for c, pair in res.items():
        d = np.abs(pair[0]["params"] - base)
        assert d.max() <= 6.1e-4, (c, d.max())
        assert (d > 2e-5).mean() <= 2e-2, (c, float((d > 2e-5).mean()))
    assert np.abs(base).max() > 0


def test_folded_layerscale_under_accumulation_and_sharded_reducer(pkg, tmp_path):
    """Folded LayerScale (raw sums -> gradients of W, b and gamma when a block's bucket leaves) against the unfolded round-4 form,
    with TWO accumulated micro-batches and the sharded reducer's collectives forced at world size 1: the same flat gradient up to
    the bf16 rounding of the folded operands.  A finish that is skipped, run twice, or run before the second micro-batch's
    sums have arrived moves the gamma / proj / fc2 gradients by far more than that."""
    helper = os.path.join(HERE, "helpers", "fold_accumulate.py")
    res = {}
    for fold in ("1", "0"):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        out = os.path.join(str(tmp_path), "fold%s.npz" % fold)
        env = dict(os.environ, VLM_FOLD_LAYERSCALE=fold, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        r = subprocess.run([sys.executable, helper, out], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        res[fold] = np.load(out)
    g1, g0 = res["1"]["grad"], res["0"]["grad"]
    names, offs = list(res["1"]["names"]), list(res["1"]["offsets"])
    assert np.isfinite(g1).all() and np.abs(g0).max() > 0
    bounds = offs + [len(g1)]
    worst = ("", 0.0)
    for i, n in enumerate(names):
        a, b = g1[bounds[i]:bounds[i + 1]], g0[bounds[i]:bounds[i + 1]]
        if np.abs(b).max() == 0:
            continue
        rel = float(np.linalg.norm(a - b) / np.linalg.norm(b))
        if rel > worst[1]:
            worst = (n, rel)
        lim = 6e-2 if ("gamma" in n or "proj" in n or "fc2" in n) else 3e-2
        assert rel <= lim, (n, rel)
    print("folded vs unfolded under accumulation, worst tensor:", worst)


def test_two_ranks_with_the_data_parallel_cu_budget(pkg, tmp_path):
    """The default a reducer takes at world size > 1 over RCCL -- GEMM grids and split-K slice counts planned for 248 of the
    256 CUs (ddp.FlatGradReducer cu_budget="auto") -- cannot be reached by the two-gloo-rank tests (the budget is for nccl
    only), so it is set by hand here (VLM_GEMM_CUS=248): both ranks must still hold BIT-identical parameters after three AdamW
    steps, and the averaged gradient must equal the single-process gradient of the concatenated batch as in the 256-CU run."""
    sys.path.insert(0, os.path.join(HERE, "helpers"))
    import ddp_one_device as H
    model, vm = H.build_model()
    nb = H.fixed_mask_batch(4)
    loss = model.training_step({"vl": H.gpu_rows(nb, 0, 4)}, 0)
    loss.backward()
    torch.cuda.synchronize()
    f = model._flat
    g_ref = f.flat_g[:f.numel].cpu().numpy().copy()
    del model
    torch.cuda.empty_cache()
    r0, r1 = run_pair(str(tmp_path), "allreduce", {"VLM_GEMM_CUS": "248"})
    assert np.array_equal(r0["params"], r1["params"]) and bool(r0["shadow_ok"]) and bool(r1["shadow_ok"])
    scale = np.abs(g_ref).max()
    assert np.abs(r0["grad0"] - g_ref).max() <= 2e-2 * scale


def test_contention_standin_leaves_the_step_unchanged(pkg):
    """The single-GPU stand-in for a collective's contention (k workgroups holding a CU each + a copy of the bucket on the
    communication stream, ddp.FlatGradReducer(standin=...)) together with the 248-CU budget: the same losses and, after three
    steps, the same parameters (to the split-K summation order the budget changes) as the plain single-process run -- the
    stand-in only occupies the machine, it must not touch the gradients."""
    import importlib
    sys.path.insert(0, os.path.join(HERE, "helpers"))
    import ddp_one_device as H
    ddp = importlib.import_module("vl_merging_amd.ddp")
    out = {}
    for tag, kw in (("plain", {}), ("standin", dict(standin=dict(cus=16), cu_budget=248))):
        model, vm = H.build_model()
        (opt,), (sch,) = vm.vilt_utils.set_schedule(model, max_steps=100)
        red = ddp.FlatGradReducer(model, **kw).attach(opt)
        nb = H.fixed_mask_batch(2)
        losses = []
        for it in range(3):
            red.begin_step()
            loss = model.training_step({"vl": H.gpu_rows(nb, 0, 2)}, it)
            loss.backward()
            red.finish_backward()
            opt.step()
            sch["scheduler"].step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        f = model._flat
        out[tag] = (losses, f.flat_p[:f.numel].cpu().numpy().copy())
        assert red.cu_budget_set == (248 if tag == "standin" else None)
        red.close()
        del model, opt, red
        torch.cuda.empty_cache()
    lib = importlib.import_module("vl_merging_amd._lib").get_lib()
    assert lib.vlm_device_cus() >= 256  # close() put the budget back
    for a, b in zip(out["plain"][0], out["standin"][0]):
        assert abs(a - b) <= 2e-3, out
    dp = np.abs(out["plain"][1] - out["standin"][1]).max()
    assert dp <= 2e-3 * max(1.0, np.abs(out["plain"][1]).max()), dp
