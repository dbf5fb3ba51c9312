"""Kernel launches of one training step by origin (this library's HIP kernels / torch's own / memcpy-memset), from torch.profiler
over 2 steps of the bench workload.  python tools/count_launches.py [arch]"""
import collections
import importlib
import sys

import torch
from torch.profiler import ProfilerActivity, profile

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
N = 2
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
OURS = ("vlm_", "attn_", "ln_", "scale_bwd", "colsum", "colreduce", "adamw", "cast_kernel", "im2col", "embedding_bwd", "bias_dense",
        "transpose_tiles", "droppath", "splitk", "merge", "l2norm", "scale_by_scalar", "contrastive", "small_ce", "ce_reduce",
        "cross_entropy", "layerscale", "gram", "potrf", "trsm", "text_rows", "image_rows", "image_lead", "fold_parts", "tanh_fwd", "act_bwd",
        "sample_negatives", "weighted_sum", "scatter_rows")
cnt, tim = collections.Counter(), collections.Counter()
names = collections.Counter()
for e in prof.events():
    if e.device_type is None or str(e.device_type).endswith("CPU"):
        continue
    n = e.name
    k = "ours" if any(o in n for o in OURS) else ("copy/fill" if ("copyBuffer" in n or "fillBuffer" in n or "Memcpy" in n or "Memset" in n) else "torch")
    cnt[k] += 1
    tim[k] += e.device_time if hasattr(e, "device_time") else e.cuda_time
    if k != "ours":
        names[n[:60]] += 1
for k in cnt:
    print("%-10s %6.1f launches per step  %8.2f ms per step" % (k, cnt[k] / N, tim[k] / N / 1e3))
print("most frequent torch / copy kernels per step:")
for n, c in names.most_common(14):
    print("  %5.1f  %s" % (c / N, n))
