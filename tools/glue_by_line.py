"""torch's own kernels inside a training step, attributed to the source line of this package that issued them."""
import collections, importlib, sys
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
cfg = cfgmod.make_config("task_mlm_itm_ifm_square_randaug_base_vl", "step200k", "ufo", image_size=384,
                         vit="vit_base_patch16_384", per_gpu_batchsize=22, num_gpus=1, vl_mlm_prob=0.25)
torch.manual_seed(0)
dev = torch.device("cuda", 0)
model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).to(dev)
model.train(); model.setup_engine()
(opt,), (sch,) = vu.set_schedule(model, max_steps=cfg["max_steps"])
red = ddp.FlatGradReducer(model).attach(opt)
batch = bench.synthetic_batch(22, 384, 40, cfg["vocab_size"], 1234, dev)
def step():
    red.begin_step(); loss = model.training_step(batch, 0); loss.backward(); red.finish_backward(); opt.step(); sch["scheduler"].step()
for _ in range(3): step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
agg = collections.Counter()
class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if "backward" not in name and any(x in name for x in ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.expand", "aten.detach", "aten.alias", "aten.slice", "aten.select", "aten.unsqueeze", "aten.squeeze", "aten.permute", "aten.as_strided", "aten.empty", "aten.reshape", "aten.unbind", "aten.split", "aten.record_stream", "aten.new_empty", "aten.is_", "aten.sym_", "aten._local_scalar", "aten.lift_fresh", "prim.")):
            return out
        where = "(autograd engine)"
        chain = []
        for fr in reversed(traceback.extract_stack()):
            if ("vl-merging_amd" in fr.filename or "vl_merging_amd" in fr.filename) :
                where = "%s:%d %s" % (fr.filename.split("/")[-1], fr.lineno, fr.name); break
        agg[(where, name.replace("aten.", ""))] += 1
        return out
with Mode():
    step()
torch.cuda.synchronize()
by_line = collections.Counter()
for (w, n), c in agg.items(): by_line[w] += c
print("dispatched non-view aten ops in one step: %d" % sum(agg.values()))
for w, c in by_line.most_common():
    ops_ = ", ".join("%s x%d" % (n, k) for (ww, n), k in sorted(agg.items(), key=lambda kv: -kv[1]) if ww == w)[:400]
    print("%4d  %-52s %s" % (c, w, ops_))
