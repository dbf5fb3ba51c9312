"""Is the folded path reproducible, and how do cls_token gradients move between B = 4 and the average of two B = 2 halves?"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/helpers")
import __graft_entry__ as ge
ge.import_package()
import ddp_gather_losses as H
eng = importlib.import_module("vl_merging_amd.engine")
config = "irtr"
for fold in (1, 0):
    os.environ["VLM_FOLD_LAYERSCALE"] = str(fold)
    eng._FOLD_LS = bool(fold)
    H.deterministic_negatives()
    model, vm = H.build_model(H.LOSSES[config], max_vl=None)
    nb = H.fixed_mask_batch(4)
    f = model._flat
    outs = []
    for rep in range(3):
        batch = H.gpu_rows(nb, 0, 4)
        f.flat_g.zero_()
        vm.vilt_utils.set_task(model)
        ret = model(H.wrap(config, dict(batch)))
        loss = sum(v for k, v in ret.items() if "loss" in k)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((f.flat_g[:f.numel].cpu().numpy().copy(), float(loss)))
    o, k = f.offsets["transformer.cls_token"]
    print("fold", fold, "losses", [x[1] for x in outs])
    print("  repeat max|diff| whole buffer:", float(np.abs(outs[0][0] - outs[1][0]).max()), float(np.abs(outs[1][0] - outs[2][0]).max()),
          " cls_token grad max:", [float(np.abs(x[0][o:o + k]).max()) for x in outs], " global max", float(np.abs(outs[0][0]).max()))
    raw = getattr(f, "_ls_buf", None)
    if raw is not None:
        print("  raw buffers after backward: max|raw| =", float(raw.abs().max()))
