import sys, json, os, importlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import __graft_entry__ as ge
ge.import_package()
import test_model_gpu as T
mods = (importlib.import_module("vl_merging_amd.vilt.config"), importlib.import_module("vl_merging_amd.vilt.modules.vilt_module"))
gd = "/root/repo/tests/golden"
for arch in ("ufo", "all_moe"):
    gold = np.load(f"{gd}/model_tiny_{arch}.npz")
    model = T.build(mods, arch, f"tiny_{arch}", gd, {"itm": 1, "mlm": 1, "ifm": 1})
    batch = T.gpu_batch(T.det_batch(2, 224, 40, 1024, seed=1234))
    model.zero_grad()
    mods[1].vilt_utils.set_task(model)
    ret = model({"vl": batch})
    total = sum(v for k, v in ret.items() if "loss" in k)
    total.backward()
    print(arch, {k: (float(ret[k]), float(gold["step/"+k])) for k in ("mlm_loss","ifm_loss","itm_loss")})
    print("ifm logits", ret["ifm_i2t_logits"].detach().cpu().numpy(), gold["step/ifm_i2t_logits"])
    gs = json.loads(str(gold["step/grad_summary"]))
    named = dict(model.named_parameters())
    rows = []
    for n, v in gs.items():
        if v is None: continue
        nrm = float(named[n].grad.double().norm())
        rows.append((abs(nrm - v[0]) / (v[0] + 1e-12), n, nrm, v[0]))
    rows.sort(reverse=True)
    for r in rows[:12]: print("%.4f %s %.5g %.5g" % r)
    for key in gold.files:
        if key.startswith("step/grad/logit"):
            n = key[len("step/grad/"):]
            print(n, float(named[n].grad), float(gold[key]))
