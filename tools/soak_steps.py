"""Soak run: N training steps of the bench workload (base_vl ufo 384^2, B = 22, train mode) on one synthetic batch; every step's loss is
kept on the device and checked at the end (all finite; the loss of a fixed batch must fall).  A cheap net for rare races: a wrong
attention row shows up as a NaN or a jump within a few hundred steps.   python tools/soak_steps.py [steps] [arch]"""
import importlib
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge

ge.import_package()
cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
vu = importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils")
syn = importlib.import_module("vl_merging_amd.synthetic")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
arch = sys.argv[2] if len(sys.argv) > 2 else "ufo"
cfg = cfgmod.make_config("task_mlm_itm_ifm_square_randaug_base_vl", "step200k", arch, image_size=384, vit="vit_base_patch16_384",
                         per_gpu_batchsize=22, num_gpus=1, vl_mlm_prob=0.25, learning_rate=2e-5, warmup_steps=20)
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).to(dev)
model.train()
model.setup_engine()
(opt,), (sch,) = vu.set_schedule(model, max_steps=cfg["max_steps"])
batch = syn.synthetic_batch(22, 384, cfg["max_text_len"], cfg["vocab_size"], 1234, dev)
losses = []
t0 = time.time()
for _ in range(steps):
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    sch["scheduler"].step()
    losses.append(loss.detach())
torch.cuda.synchronize()
dt = time.time() - t0
l = torch.stack(losses).float().cpu()
print("steps %d in %.1f s (%.1f ms per step); loss first %.4f min %.4f last %.4f; finite: %s" %
      (steps, dt, dt / steps * 1e3, float(l[0]), float(l.min()), float(l[-1]), bool(torch.isfinite(l).all())))
assert bool(torch.isfinite(l).all()), "non-finite loss at steps %s" % torch.nonzero(~torch.isfinite(l)).flatten().tolist()[:10]
assert float(l[-20:].mean()) < float(l[:20].mean()), "the loss of a fixed batch did not fall"
jumps = (l[1:] - l[:-1]).abs()
print("largest step-to-step change %.4f at step %d" % (float(jumps.max()), int(jumps.argmax()) + 1))
