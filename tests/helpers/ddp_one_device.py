"""Child process of tests/test_ddp_gpu.py: ONE rank of a 2-rank gloo group, both ranks on cuda:0 (RCCL refuses two
ranks per device; gloo stages CUDA tensors through the host).  Runs the real model through the real reducer /
optimizer in the requested configuration and saves what the parent compares.

usage: ddp_one_device.py OUTDIR CONFIG      CONFIG in {allreduce, allreduce_defer, sharded, sharded_defer}
env:   RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT
"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
from oracle.detweights import det_array, det_batch  # noqa: E402


def fixed_mask_batch(B, k=3, seed=4242):
    """det_batch with exactly k masked tokens per sample, so that the mean MLM loss of a concatenated batch equals the
    mean of its halves' means (what DDP's gradient averaging assumes)."""
    nb = det_batch(B, 224, 40, 1024, seed=seed)
    ids, labels, ids_mlm = nb["text_ids"], np.full_like(nb["text_labels_mlm"], -100), nb["text_ids"].copy()
    for b in range(B):
        ln = int(nb["text_masks"][b].sum())
        pos = 1 + (np.arange(k) * 2) % max(1, ln - 2)
        labels[b, pos] = ids[b, pos]
        ids_mlm[b, pos] = 103
    nb["text_labels_mlm"], nb["text_ids_mlm"] = labels, ids_mlm
    return nb


def build_model(train=False):
    ge.import_package()
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    cfg = cfgmod.make_config("ufo", vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, vocab_size=1024,
                             max_text_len=40, patch_size=16, vlffn_start_layer_index=10, image_size=224,
                             max_vl_text_len=40, tasks=["vl"], loss_names=cfgmod._loss_names({"mlm": 1}), warmup_steps=0)
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    sd = {k: torch.from_numpy(det_array(k, v.shape)) for k, v in model.state_dict().items()
          if v.is_floating_point() and "index" not in k and "mask_for" not in k}
    model.load_state_dict(sd, strict=False)
    model = model.cuda()
    model.train(train)
    model.setup_engine()
    return model, vm


def gpu_rows(nb, lo, hi):
    b = {k: torch.from_numpy(v[lo:hi]).cuda() for k, v in nb.items()}
    b["image"] = [b["image"]]
    return b


def main():
    outdir, config = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ddp = importlib.import_module("vl_merging_amd.ddp") if ge.import_package() else None
    model, vm = build_model()
    vu = vm.vilt_utils
    per = 2
    nb = fixed_mask_batch(per * world)
    batch = gpu_rows(nb, rank * per, (rank + 1) * per)
    (opt,), (sch,) = vu.set_schedule(model, max_steps=100)
    red = ddp.FlatGradReducer(model, sharded=config.startswith("sharded")).attach(opt, defer_tail=config.endswith("defer"))
    out = {}
    for it in range(3):
        red.begin_step()
        loss = model.training_step({"vl": batch}, it)
        loss.backward()
        red.finish_backward()
        if it == 0:
            if red.defer_tail:
                red.wait_tail()
            torch.cuda.synchronize()
            g = model._flat.flat_g[:model._flat.numel].clone()
            if red.sharded:  # only the rank's own chunk of every bucket holds the reduced sum
                keep = torch.zeros_like(g, dtype=torch.bool)
                for lo, hi in red.own_ranges():
                    keep[lo:hi] = True
                g = torch.where(keep, g, torch.zeros_like(g))
                out["own"] = keep.cpu().numpy()
            out["grad0"] = (g * red.grad_scale).cpu().numpy()
            out["loss0"] = np.array(float(loss))
        opt.step()
        sch["scheduler"].step()
    torch.cuda.synchronize()
    f = model._flat
    out["params"] = f.flat_p[:f.numel].cpu().numpy()
    out["shadow_ok"] = np.array(bool(torch.equal(f.flat_b[:f.numel], f.shadow_reference())))
    out["state_elements"] = np.array(opt.state_elements())
    out["numel"] = np.array(f.numel)
    np.savez(os.path.join(outdir, "%s_rank%d.npz" % (config, rank)), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
