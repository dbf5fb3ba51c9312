"""configs[4]'s per-GPU step on its own (task_finetune_irtr_coco 384^2, ufo, B = 20, fwd + bwd + AdamW) for a kernel trace:
python tools/irtr_step.py [steps]"""
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
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
cfg = cfgmod.make_config("task_finetune_irtr_coco_square_randaug_base_image384", "ufo", image_size=384, vit="vit_base_patch16_384",
                         per_gpu_batchsize=20, num_gpus=1)
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).to(dev)
model.train()
model.setup_engine()
(opt,), (sch,) = vu.set_schedule(model, max_steps=1000)
batch = syn.synthetic_batch(20, 384, cfg["max_text_len"], cfg["vocab_size"], 1234, dev)["vl"]


def step():
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    sch["scheduler"].step()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print("irtr ufo B=20: %.2f ms per step, %.1f samples/s" % (dt * 1e3, 20 / dt))
