"""Which torch ops (not our HIP kernels) run inside a training step, by device time: torch.profiler over 2 steps of the
bench workload.  python tools/torch_ops_profile.py [arch]"""
import importlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
import bench

ge.import_package()
cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
vu = importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils")
ddp = importlib.import_module("vl_merging_amd.ddp")
arch = sys.argv[1] if len(sys.argv) > 1 else "ufo"
cfg = cfgmod.make_config("task_mlm_itm_ifm_square_randaug_base_vl", "step200k", arch, image_size=384,
                         vit="vit_base_patch16_384", per_gpu_batchsize=22, num_gpus=1, vl_mlm_prob=0.25)
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).to(dev)
model.train()
model.setup_engine()
(opt,), (sch,) = vu.set_schedule(model, max_steps=cfg["max_steps"])
red = ddp.FlatGradReducer(model).attach(opt)
batch = bench.synthetic_batch(22, 384, 40, cfg["vocab_size"], 1234, dev)


def step():
    red.begin_step()
    loss = model.training_step(batch, 0)
    loss.backward()
    red.finish_backward()
    opt.step()
    sch["scheduler"].step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False, record_shapes=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
ev = [e for e in ev if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows = sorted(ev, key=lambda e: (-e.count if os.environ.get("VLM_PROFILE_BY_COUNT") else 0, -e.self_device_time_total))
tot = sum(e.self_device_time_total for e in ev)
print("total device time %.2f ms over 2 steps" % (tot / 1e3))
import os
flt = os.environ.get("VLM_PROFILE_FILTER")
if flt:
    import re as _re
    rows = [e for e in rows if _re.search(flt, e.key)]
for e in rows[:90]:
    print("%-28s n=%4d  dev %8.1f us  %s" % (e.key[:28], e.count, e.self_device_time_total, str(e.input_shapes)[:110]))
