#!/usr/bin/env python
"""`python run.py with <named configs> key=value ...` -- the reference's entry point (src/run.py:141-295) for the
hot path: builds the model from the sacred-style config and runs training steps on synthetic batches
(the data modules of the reference are outside the hot path).

Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 run.py with ...`.
"""
import importlib
import os
import sys
import time

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main(argv):
    ge.import_package()
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    vu = importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils")
    ddp = importlib.import_module("vl_merging_amd.ddp")
    sys.path.insert(0, ROOT)
    from bench import synthetic_batch
    steps = 10
    rest = []
    for a in argv:
        if a.startswith("steps="):
            steps = int(a.split("=", 1)[1])
        else:
            rest.append(a)
    cfg = cfgmod.parse_cli(rest)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.manual_seed(cfg["seed"])
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).cuda()
    model.train()
    model.setup_engine()
    (opt,), (sch,) = vu.set_schedule(model, max_steps=cfg["max_steps"] or 100000)
    # use_sharded_training=True: the reference's `ddp_sharded` plugin (run.py:231-232) -> reduce-scatter + sharded AdamW
    red = ddp.FlatGradReducer(model, sharded=bool(cfg.get("use_sharded_training"))).attach(opt)  # same set-up as bench.py
    B = cfg["per_gpu_batchsize"] or 2
    # run.py:155-158 of the reference: accumulate_grad_batches = batch_size // (per_gpu_batchsize * gpus * nodes)
    grad_steps = max(1, int(cfg["batch_size"]) // (B * world * max(1, int(cfg["num_nodes"]))))
    if os.environ.get("VLM_GRAD_STEPS"):
        grad_steps = int(os.environ["VLM_GRAD_STEPS"])
    opt.grad_scale = red.grad_scale / grad_steps  # mean over micro-batches and ranks, folded into AdamW
    batch = synthetic_batch(B, cfg["image_size"], cfg["max_text_len"], cfg["vocab_size"], 1234 + rank, "cuda")
    if cfg["tasks"] is None:
        batch = batch["vl"]
    if rank == 0:
        print("global batch %d = %d per GPU x %d ranks x %d accumulated micro-batches" % (B * world * grad_steps, B, world, grad_steps),
              flush=True)
    for it in range(steps):
        t0 = time.time()
        red.begin_step()
        for micro in range(grad_steps):
            red.accumulate = micro + 1 < grad_steps  # only the last micro-batch's backward sends gradients
            if micro:
                for i in red.seen:
                    red.seen[i] = 0
            loss = model.training_step(batch, it)
            loss.backward()
            red.finish_backward()
        opt.step()
        sch["scheduler"].step()
        if rank == 0:
            print("step %d loss %.4f  %.1f ms" % (it, float(loss.detach()), (time.time() - t0) * 1e3), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
