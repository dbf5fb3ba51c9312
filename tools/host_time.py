"""Host-side enqueue time of a training step against its GPU time: how far the launch thread runs ahead of the device
(python tools/host_time.py [ufo|all_moe]).  The step is GPU-bound while enqueue < step time."""
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
arch = sys.argv[1] if len(sys.argv) > 1 else "ufo"
cfg = cfgmod.make_config("task_mlm_itm_ifm_square_randaug_base_vl", "step200k", arch, image_size=384, vit="vit_base_patch16_384",
                         per_gpu_batchsize=22, num_gpus=1, vl_mlm_prob=0.25)
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).to(dev)
model.train()
model.setup_engine()
(opt,), (sch,) = vu.set_schedule(model, max_steps=cfg["max_steps"])
batch = syn.synthetic_batch(22, 384, cfg["max_text_len"], cfg["vocab_size"], 1234, dev)


def step():
    loss = model.training_step(batch, 0)
    loss.backward()
    opt.step()
    sch["scheduler"].step()


for _ in range(3):
    step()
torch.cuda.synchronize()
host, total = [], []
for _ in range(6):
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((t1 - t0) * 1e3)
    total.append((t2 - t0) * 1e3)
print("%s: host enqueue %.1f ms (min %.1f), step %.1f ms per step" % (arch, sorted(host)[3], min(host), sorted(total)[3]))
