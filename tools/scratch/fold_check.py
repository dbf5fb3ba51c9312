"""Folded LayerScale against the unfolded path on ONE process and ONE batch: per-parameter gradient differences."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests/helpers")
import __graft_entry__ as ge
ge.import_package()
import ddp_gather_losses as H

config = sys.argv[1] if len(sys.argv) > 1 else "irtr"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
eng = importlib.import_module("vl_merging_amd.engine")
res = {}
for fold in (1, 0):
    os.environ["VLM_FOLD_LAYERSCALE"] = str(fold)
    eng._FOLD_LS = bool(fold)
    H.deterministic_negatives()
    model, vm = H.build_model(H.LOSSES[config], max_vl=40 if config == "pretrain" else None)
    nb = H.fixed_mask_batch(B)
    batch = H.gpu_rows(nb, 0, B)
    f = model._flat
    f.flat_g.zero_()
    vm.vilt_utils.set_task(model)
    ret = model(H.wrap(config, dict(batch)))
    loss = sum(v for k, v in ret.items() if "loss" in k)
    loss.backward()
    torch.cuda.synchronize()
    res[fold] = ({n: p.grad.detach().float().cpu().numpy().copy() for n, p in model.named_parameters()}, float(loss))
    print("fold", fold, "loss", float(loss), "folded jobs", len(getattr(f, "_ls_jobs", {})))
g1, g0 = res[1][0], res[0][0]
rows = []
for n in g1:
    a, b = g1[n], g0[n]
    mx = np.abs(b).max()
    if mx == 0 and np.abs(a).max() == 0:
        continue
    rows.append((np.abs(a - b).max() / (mx + 1e-30), n, mx, np.abs(a - b).max()))
rows.sort(reverse=True)
for r in rows[:25]:
    print("%.4f  %-60s max %.3e  absdiff %.3e" % r)
