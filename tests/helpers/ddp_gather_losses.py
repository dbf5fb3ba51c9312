"""Child process of tests/test_ddp_losses_gpu.py: ONE rank of a 2-rank gloo group (both ranks on cuda:0) running the
losses that GATHER across ranks on the real model: compute_ifm / compute_itm_hardneg (through the fused 4B pass and the
asynchronous candidate prefetch) and compute_irtr.  Reference: objectives.py:176-178 (ids, masks, images of every
rank), :274-300 and :393-394 (features), pytorch_lightning DDP's gradient average (run.py:263-288).

usage: ddp_gather_losses.py OUTDIR CONFIG      CONFIG in {pretrain, irtr}
env:   RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT

Hard negatives are made deterministic for the comparison with one process on the concatenated batch: torch.multinomial
is replaced by an argmax over the same weights (the candidate SET is rank-independent, its order is not).
"""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import __graft_entry__ as ge  # noqa: E402
from oracle.detweights import det_array  # noqa: E402
from ddp_one_device import fixed_mask_batch, gpu_rows  # noqa: E402

PER = 2  # samples per rank (B * W >= 2 is what compute_itm_hardneg needs, objectives.py:197-205)


def deterministic_negatives():
    torch.multinomial = lambda w, n, *a, **k: w.argmax(dim=1, keepdim=True)


def build_model(losses, max_vl=40):
    ge.import_package()
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    cfg = cfgmod.make_config("ufo", vit="vit_tiny_patch16_224", hidden_size=192, num_heads=3, vocab_size=1024,
                             max_text_len=40, patch_size=16, vlffn_start_layer_index=10, image_size=224,
                             max_vl_text_len=max_vl, tasks=["vl"] if max_vl else None,
                             loss_names=cfgmod._loss_names(losses), warmup_steps=0)
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    sd = {k: torch.from_numpy(det_array(k, v.shape)) for k, v in model.state_dict().items()
          if v.is_floating_point() and "index" not in k and "mask_for" not in k}
    model.load_state_dict(sd, strict=False)
    model = model.cuda()
    model.train(False)  # no DropPath / dropout draws: the two set-ups must see the same network
    model.setup_engine()
    return model, vm


LOSSES = {"pretrain": {"mlm": 1, "itm": 1, "ifm": 1}, "irtr": {"irtr": 1, "itm": 0, "mlm": 0, "ifm": 0}}


def wrap(config, batch):
    return {"vl": batch} if config == "pretrain" else batch


def main():
    outdir, config = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    deterministic_negatives()
    model, vm = build_model(LOSSES[config], max_vl=40 if config == "pretrain" else None)
    ddp = importlib.import_module("vl_merging_amd.ddp")
    obj = importlib.import_module("vl_merging_amd.vilt.modules.objectives")
    vu = vm.vilt_utils
    nb = fixed_mask_batch(PER * world)
    batch = gpu_rows(nb, rank * PER, (rank + 1) * PER)
    out = {}
    # (i) what the gathers return: own block first, then the other ranks in rank order (objectives.py:269-286)
    pre = obj._CandidatePrefetch(batch)
    ids, masks, images = pre.result()
    out["cand_text_ids"], out["cand_text_masks"] = ids.cpu().numpy(), masks.cpu().numpy()
    out["cand_images"] = images.cpu().numpy()
    out["cand_images_plain"] = obj._gather_cat(batch["image"][0]).cpu().numpy()  # the fp32 gather of the reference
    tag = torch.full((PER, 4), float(rank + 1), device="cuda").requires_grad_(True)
    got = obj._gather_first_own(tag)
    out["first_own"] = got.detach().cpu().numpy()
    got.sum().backward()
    out["first_own_grad"] = tag.grad.cpu().numpy()  # gradients flow through the local slice only
    # (iii) one step through the reducer
    (opt,), _ = vu.set_schedule(model, max_steps=100)
    red = ddp.FlatGradReducer(model).attach(opt, defer_tail=False)
    red.begin_step()
    vu.set_task(model)
    ret = model(wrap(config, batch))
    loss = sum(v for k, v in ret.items() if "loss" in k)
    loss.backward()
    red.finish_backward()
    torch.cuda.synchronize()
    f = model._flat
    out["grad_avg"] = (f.flat_g[:f.numel] * red.grad_scale).cpu().numpy()
    for k, v in ret.items():
        if "loss" in k:
            out[k] = np.array(float(v.detach()))
    opt.step()
    torch.cuda.synchronize()
    out["params"] = f.flat_p[:f.numel].cpu().numpy()
    np.savez(os.path.join(outdir, "%s_rank%d.npz" % (config, rank)), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
