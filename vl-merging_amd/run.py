#!/usr/bin/env python
"""`python run.py with <named configs> key=value ...` -- the reference's entry point (src/run.py:141-295) on the MI355X
engine: config -> data -> model (checkpoint load / merge in __init__) -> fit | validate | test.

What the reference delegates to pytorch_lightning is restated for the hot path only:
  * data (run.py:160-163): `data_root=<dir of Arrow shards + vocab.txt>` -> vilt.datamodules.ArrowBatches (the reference's
    shard names per dataset and split, DistributedSampler semantics, the MLM collator); without a data_root the step runs
    on one synthetic batch per rank (bench.py's workload);
  * fit (run.py:295): gradient accumulation `batch_size // (per_gpu_batchsize * gpus * nodes)` (:210-212), fused AdamW +
    polynomial schedule, gradient all-reduce overlapped with backward (ddp.FlatGradReducer; `use_sharded_training` ->
    the sharded optimizer, :231-232), `max_steps` / `max_epoch`, `limit_train_batches`;
  * ModelCheckpoint(save_last=True) (run.py:189-195): `<log_dir>/<exp>_seed<seed>_from_<ckpt>/version_<n>/checkpoints/
    last.ckpt`, a Lightning-layout pickle (state_dict, optimizer_states, lr_schedulers, global_step, epoch,
    hyper_parameters) that the reference's own `load_path=` / `resume_from_checkpoint` read;
  * resume (run.py:218-223, :280): `resume_during_pretraining=True` picks the last `version_*/checkpoints/last.ckpt` of the
    run directory; `resume_from=<path>` names one;
  * validation_only / test_only (run.py:290-293): the task's losses over the val / test split; with `get_recall_metric`
    the retrieval recalls of compute_irtr_recall (vilt_utils.py:66-99 of the reference).
Extra CLI words: `steps=N` (stop after N optimizer steps whatever the config says).

Multi-GPU: `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 run.py with ...`.
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


def run_dir(cfg):
    """TensorBoardLogger(log_dir, name=f"{exp}_seed{seed}_from_{ckpt stem}") of run.py:196-199."""
    stem = cfg["load_path"].split("/")[-1][:-5]
    return os.path.join(cfg["log_dir"], "%s_seed%s_from_%s" % (cfg["exp_name"], cfg["seed"], stem))


def find_resume(cfg):
    """run.py:218-223: the LAST existing version_<i>/checkpoints/last.ckpt, i in 0..99."""
    found = None
    for i in range(100):
        p = os.path.join(run_dir(cfg), "version_%d" % i, "checkpoints", "last.ckpt")
        if os.path.exists(p):
            found = p
    return found


def next_version_dir(cfg):
    base, i = run_dir(cfg), 0
    while os.path.exists(os.path.join(base, "version_%d" % i)):
        i += 1
    return os.path.join(base, "version_%d" % i)


def log(rank, *a):
    if rank == 0:
        print(*a, flush=True)


def evaluate(model, cfg, dm_mod, vu, obj, split, rank, world, dev):
    """trainer.validate / trainer.test for the hot-path tasks: mean losses over the split (ranks take batches round
    robin; sums meet by all-reduce) and, with get_recall_metric, the retrieval recalls."""
    model.eval()
    vu.set_task(model)
    out = {}
    data = dm_mod.ArrowBatches(cfg, split, rank, world)
    sums, n = {}, 0
    with torch.no_grad(), obj.local_only():
        for batch in data.eval_batches(dev):
            ret = model({"vl": batch} if cfg["tasks"] is not None else batch)
            bsz = batch["text_ids"].shape[0]
            for k, v in ret.items():
                if "loss" in k:
                    sums[k] = sums.get(k, 0.0) + float(v) * bsz
            n += bsz
    keys = sorted(sums)
    if world > 1:  # a rank that was dealt no batch (more ranks than batches) has seen no loss: agree on the key set first
        every = [None] * world
        dist.all_gather_object(every, keys)
        keys = sorted(set(k for ks in every for k in ks))
    t = torch.tensor([sums.get(k, 0.0) for k in keys] + [float(n)], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
    for i, k in enumerate(keys):
        out["%s/%s" % (split, k)] = float(t[i] / max(1.0, float(t[-1])))
    out["%s/samples" % split] = int(t[-1])
    if cfg["get_recall_metric"]:
        txt = dm_mod.ArrowBatches(cfg, split, rank, world, image_only=False, tokenizer=data.tokenizer)
        img = dm_mod.ArrowBatches(cfg, split, rank, world, image_only=True, tokenizer=data.tokenizer)
        text_preload = [{"text_ids": b["text_ids"], "text_masks": b["text_masks"], "text_labels": b["text_labels"],
                         "img_index": b["img_index"]} for b in txt.eval_batches("cpu", all_ranks=True)]
        T = cfg["max_text_len"]
        image_preload = [{"image": b["image"], "img_index": b["img_index"],
                          "text_masks": torch.ones(len(b["img_index"]), T, dtype=torch.long)}
                         for b in img.eval_batches("cpu", all_ranks=True)]
        r = obj.compute_irtr_recall(model, text_preload, image_preload)
        for name, v in zip(("ir_r1", "ir_r5", "ir_r10", "tr_r1", "tr_r5", "tr_r10"), r[:6]):
            out["recalls/" + name] = float(v)
    return out


def main(argv):
    ge.import_package()
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    vu = importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils")
    obj = importlib.import_module("vl_merging_amd.vilt.modules.objectives")
    ddp = importlib.import_module("vl_merging_amd.ddp")
    ckpt_mod = importlib.import_module("vl_merging_amd.checkpoint")
    dm_mod = importlib.import_module("vl_merging_amd.vilt.datamodules")
    synthetic_batch = importlib.import_module("vl_merging_amd.synthetic").synthetic_batch
    steps_cap = None
    rest = []
    for a in argv:
        if a.startswith("steps="):
            steps_cap = int(a.split("=", 1)[1])
        else:
            rest.append(a)
    cfg = cfgmod.parse_cli(rest)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0)) if os.environ.get("VLM_BENCH_ONE_DEVICE", "0") == "0" else 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29579")
        dist.init_process_group(os.environ.get("VLM_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    torch.manual_seed(cfg["seed"])  # pl.seed_everything, run.py:145

    # ---- resume (run.py:218-223): the checkpoint replaces load_path's weights AFTER the model is built ------------------
    resume = cfg["resume_from"] or (find_resume(cfg) if cfg["resume_during_pretraining"] else None)
    log(rank, "resume_ckpt: %s" % resume)
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg))
    resumed = None
    if resume:
        resumed = ckpt_mod.load_file(resume)
        info = model.load_state_dict(resumed["state_dict"], strict=False)
        log(rank, "resumed weights: %d missing, %d unexpected keys" % (len(info.missing_keys), len(info.unexpected_keys)))
    model = model.cuda()
    model.setup_engine()

    if cfg["validation_only"] or cfg["test_only"]:  # run.py:290-293
        if not cfg["data_root"]:
            raise ValueError("validation_only / test_only need data_root=<dir of Arrow shards>")
        res = evaluate(model, cfg, dm_mod, vu, obj, "test" if cfg["test_only"] else "val", rank, world, dev)
        for k in sorted(res):
            log(rank, "%s %s" % (k, res[k]))
        if world > 1:
            dist.destroy_process_group()
        return res

    # ---- fit -------------------------------------------------------------------------------------------------------
    model.train()
    B = cfg["per_gpu_batchsize"] or 2
    grad_steps = max(1, int(cfg["batch_size"]) // (B * world * max(1, int(cfg["num_nodes"]))))  # run.py:210-212
    if os.environ.get("VLM_GRAD_STEPS"):
        grad_steps = int(os.environ["VLM_GRAD_STEPS"])
    data = dm_mod.ArrowBatches(cfg, "train", rank, world) if cfg["data_root"] else None
    if data is not None:
        per_epoch = data.steps_per_epoch()
        lim = cfg["limit_train_batches"]  # Lightning: a float is a fraction of the epoch, an int a number of batches
        if isinstance(lim, float) and lim < 1.0:
            per_epoch = int(per_epoch * lim)
        elif isinstance(lim, int) and not isinstance(lim, bool) and lim >= 1:
            per_epoch = min(per_epoch, lim)
        if per_epoch < grad_steps:
            raise ValueError("the train split gives %d batches per rank and epoch, fewer than the %d accumulated per step"
                             % (per_epoch, grad_steps))
    max_steps = cfg["max_steps"]
    if max_steps is None:  # vilt_utils.py:323-330 of the reference: from the dataloader length
        if data is None:
            max_steps = 100000
        else:
            max_steps = per_epoch * int(cfg["max_epoch"]) // grad_steps  # len(loader) * max_epochs // accumulate_grad_batches
    (opt,), (sch,) = vu.set_schedule(model, max_steps=max_steps)
    red = ddp.FlatGradReducer(model, sharded=bool(cfg.get("use_sharded_training"))).attach(opt)  # same set-up as bench.py
    opt.grad_scale = red.grad_scale / grad_steps  # mean over micro-batches and ranks, folded into AdamW
    global_step, epoch, in_epoch = 0, 0, 0  # in_epoch: micro-batches of the current epoch consumed so far
    if resumed is not None:
        # the position of the run (step, epoch, schedule, place in the epoch) is restored whether or not the file holds Adam's
        # moments: a weights-only checkpoint must not restart the warm-up at step 0 on trained weights
        global_step, epoch = int(resumed.get("global_step", 0)), int(resumed.get("epoch", 0))
        in_epoch = int(resumed.get("vlm_micro_batches_in_epoch", 0))
        if resumed.get("lr_schedulers"):
            sch["scheduler"].load_state_dict(resumed["lr_schedulers"][0])
        else:
            sch["scheduler"].load_state_dict({"last_epoch": global_step})
        if resumed.get("optimizer_states"):
            opt.load_state_dict(resumed["optimizer_states"][0])
            log(rank, "resumed optimizer at global_step %d, epoch %d (+%d micro-batches)" % (global_step, epoch, in_epoch))
        else:
            opt.step_count = global_step  # Adam's bias correction continues; its moments restart from zero
            log(rank, "WARNING: %s holds no optimizer_states: Adam's moments restart from zero at global_step %d (schedule, "
                      "epoch and data position restored)" % (resume, global_step))
    last_step = max_steps if steps_cap is None else min(max_steps, global_step + steps_cap)
    log(rank, "global batch %d = %d per GPU x %d ranks x %d accumulated micro-batches; steps %d -> %d"
        % (B * world * grad_steps, B, world, grad_steps, global_step, last_step))

    def micro_batches():
        """Endless stream of this rank's micro-batches: epochs of the train split, or the one synthetic batch."""
        nonlocal epoch, in_epoch
        if data is None:
            b = synthetic_batch(B, cfg["image_size"], cfg["max_text_len"], cfg["vocab_size"], 1234 + rank, dev,
                                loss_names=cfg["loss_names"], vqav2_label_size=cfg["vqav2_label_size"])
            b = b if cfg["tasks"] is not None else b["vl"]
            while True:
                yield b
        while True:
            for b in data.train_epoch(epoch, dev, skip=in_epoch):
                if in_epoch >= per_epoch:
                    break
                in_epoch += 1
                seen.append(list(b["raw_index"]))
                yield {"vl": b} if cfg["tasks"] is not None else b
            epoch += 1
            in_epoch = 0

    # ---- ModelCheckpoint(save_last=True) (run.py:189-195): last.ckpt at every validation interval and at the end ---------------
    vdir = None
    if cfg["log_dir"]:
        vdir = os.path.dirname(os.path.dirname(resume)) if resume and os.path.dirname(resume).endswith("checkpoints") \
            and os.path.abspath(resume).startswith(os.path.abspath(run_dir(cfg))) else next_version_dir(cfg)
    vci = cfg["val_check_interval"]  # Lightning: a float is a fraction of the epoch, an int a number of training batches
    if os.environ.get("VLM_SAVE_EVERY"):
        save_every = int(os.environ["VLM_SAVE_EVERY"])
    elif data is None:
        save_every = 0
    elif isinstance(vci, float):
        save_every = max(1, int(per_epoch * vci) // grad_steps)
    else:
        save_every = max(1, int(vci) // grad_steps)

    def save_last():
        """Every rank calls it (a sharded optimizer's state is gathered collectively); rank 0 writes, through a temporary
        file and os.replace: a crash mid-save leaves the previous last.ckpt (the resume source) intact."""
        extra = {"lr_schedulers": [sch["scheduler"].state_dict()], "vlm_micro_batches_in_epoch": in_epoch,
                 "optimizer_states": [opt.state_dict()] if (rank == 0 or red.sharded) else None}
        if rank != 0 or vdir is None:
            return None
        os.makedirs(os.path.join(vdir, "checkpoints"), exist_ok=True)
        path = os.path.join(vdir, "checkpoints", "last.ckpt")
        tmp = path + ".tmp.%d" % os.getpid()
        ckpt_mod.save_ckpt(tmp, model, global_step=global_step, epoch=epoch, extra=extra)
        os.replace(tmp, path)
        print("saved %s (global_step %d)" % (path, global_step), flush=True)
        return path

    seen = []  # dataset indices of every micro-batch this rank consumed (returned: tests check the resumed data order)
    stream = micro_batches()
    loss = None
    path = None
    saved_at = -1
    while global_step < last_step:
        t0 = time.time()
        red.begin_step()
        for micro in range(grad_steps):
            red.accumulate = micro + 1 < grad_steps  # only the last micro-batch's backward sends gradients
            if micro:
                for i in red.seen:
                    red.seen[i] = 0
            loss = model.training_step(next(stream), global_step)
            loss.backward()
            red.finish_backward()
        opt.step()
        sch["scheduler"].step()
        global_step += 1
        if rank == 0 and (global_step % 10 == 0 or global_step == last_step or global_step <= 3):
            print("step %d loss %.4f  %.1f ms" % (global_step, float(loss.detach()), (time.time() - t0) * 1e3), flush=True)
        if save_every and global_step % save_every == 0 and global_step < last_step:
            path = save_last() or path
            saved_at = global_step

    if world > 1:
        dist.barrier()
    if saved_at != global_step:
        path = save_last() or path
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return {"last_ckpt": path, "global_step": global_step, "loss": float(loss.detach()) if loss is not None else None,
            "seen_raw_index": seen, "epoch": epoch}


if __name__ == "__main__":
    main(sys.argv[1:])
