#!/usr/bin/env python
"""`python cache_gram_matrices.py with <named configs> key=value ... representation_name=<name>`
-- the reference's Gram-matrix cache (src/cache_gram_matrices.py:141-357) on the MI355X engine.

The reference registers forward hooks that add X^T X (float64) of the input of every linear / attention module and
drives them with `trainer.validate` over the retrieval validation set; here the fused block function feeds the same
inputs to an on-device float64 accumulator (v_mfma_f64 SYRK, ONE device->host copy at the end instead of one 4.7/75 MB
copy per hook call).  Batches come from Arrow shards (`data_root=<dir>`: ArrowDataset over every *.arrow file in it,
vocab.txt next to them) or, without a data_root, from synthetic COCO-shaped batches.  Multi-GPU: launch with
torch.distributed.run; the batches are dealt round-robin over the ranks and the Gram sums are all-reduced.  Output: `<log_dir>/<representation_name>.pth` = torch.save(dict name -> float64 [D,D]),
the file `regmean` loads at vilt_module.py:386.
"""
import importlib
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def arrow_batches(cfg, B, dev, rank=0, world=1, limit=None):
    """Validation batches of the retrieval task from Arrow shards under cfg["data_root"] (the reference drives its hooks
    with trainer.validate over the same shards, cache_gram_matrices.py:339): ArrowDataset + its collate, a
    non-masking collator (evaluation reads the plain ids).  EVERY sample is hooked, as under trainer.validate: the last
    batch may be ragged.  Batch j belongs to rank j % world; a rank decodes and collates only its own batches."""
    ds = importlib.import_module("vl_merging_amd.vilt.datasets")
    names = sorted(f[:-6] for f in os.listdir(cfg["data_root"]) if f.endswith(".arrow"))
    vocab = os.path.join(cfg["data_root"], "vocab.txt")
    if not os.path.isfile(vocab):
        raise FileNotFoundError("cache_gram_matrices: %s (tokenizer vocabulary next to the shards)" % vocab)
    tok = ds.build_synthetic_tokenizer(vocab) if os.path.getsize(vocab) < 4096 else None
    if tok is None:
        from transformers import BertTokenizer
        tok = BertTokenizer(vocab, do_lower_case=True)
    data = ds.ArrowDataset(cfg["data_root"], ["square_transform"], cfg["image_size"], names, text_column_name="caption",
                           max_text_len=cfg["max_text_len"], tokenizer=tok)

    def no_mask(encodings):
        ids = torch.zeros(len(encodings), cfg["max_text_len"], dtype=torch.long)
        for r, e in enumerate(encodings):
            ids[r, : len(e["input_ids"])] = torch.tensor(e["input_ids"])
        return {"input_ids": ids.clone(), "labels": torch.full_like(ids, -100)}

    starts = list(range(0, len(data), B))
    if limit is not None:
        starts = starts[:limit]
    for j, lo in enumerate(starts):
        if j % world != rank:
            continue
        b = data.collate([data[i] for i in range(lo, min(lo + B, len(data)))], no_mask)
        yield {"image": [b["image"][0].to(dev)], "text_ids": b["text_ids"].to(dev), "text_masks": b["text_masks"].to(dev),
               "text_labels": b["text_labels"].to(dev), "text_ids_mlm": b["text_ids_mlm"].to(dev),
               "text_labels_mlm": b["text_labels_mlm"].to(dev)}


def main(argv):
    import torch.distributed as dist
    ge.import_package()
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    vu = importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils")
    synthetic_batch = importlib.import_module("vl_merging_amd.synthetic").synthetic_batch
    batches = None  # batches=N: stop after N batches (all ranks together); default: 4 synthetic batches / every Arrow sample
    rest = []
    for a in argv:
        if a.startswith("batches="):
            batches = int(a.split("=", 1)[1])
        else:
            rest.append(a)
    cfg = cfgmod.parse_cli(rest)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0)) if os.environ.get("VLM_BENCH_ONE_DEVICE", "0") == "0" else 0
    torch.cuda.set_device(local)
    if world > 1:  # one process per GPU; every rank hooks its shard of the batches, the float64 sums meet over RCCL
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29578")
        dist.init_process_group(os.environ.get("VLM_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    torch.manual_seed(cfg["seed"])
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).cuda().eval()
    model.setup_engine()
    vu.set_task(model)
    cap = model.start_gram_capture()
    B = cfg["per_gpu_batchsize"] or 2
    if cfg["data_root"]:
        source = arrow_batches(cfg, B, "cuda", rank, world, batches)
    else:  # DistributedSampler-style round robin over the batches: batch i is built and hooked by rank i % world only
        source = (synthetic_batch(B, cfg["image_size"], cfg["max_text_len"], cfg["vocab_size"], 4321 + i, "cuda")["vl"]
                  for i in range(4 if batches is None else batches) if i % world == rank)
    n_seen = 0
    obj = importlib.import_module("vl_merging_amd.vilt.modules.objectives")
    # the ranks run different numbers of forward passes (round robin over a ragged batch list) and only the hooks matter:
    # the losses' cross-rank gathers are switched off for the sweep
    with torch.no_grad(), obj.local_only():
        for batch in source:
            model(batch)
            n_seen += 1
    model.stop_gram_capture()
    cap.all_reduce()
    grams = cap.state_dict()
    if rank == 0:
        os.makedirs(cfg["log_dir"], exist_ok=True)
        path = os.path.join(cfg["log_dir"], cfg["representation_name"] + ".pth")
        torch.save(grams, path)
        for k, v in list(grams.items())[:4]:
            print(k, tuple(v.shape), float(v.min()), float(v.max()))
        print("saved %d gram matrices (%d batches on rank 0, %d ranks) to %s" % (len(grams), n_seen, world, path))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
