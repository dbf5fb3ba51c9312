#!/usr/bin/env python
"""`python cache_gram_matrices.py with <named configs> key=value ... representation_name=<name>`
-- the reference's Gram-matrix cache (src/cache_gram_matrices.py:141-357) on the MI355X engine.

The reference registers forward hooks that add X^T X (float64) of the input of every linear / attention module and
drives them with `trainer.validate` over the retrieval validation set; here the fused block function feeds the same
inputs to an on-device accumulator (MFMA product + float64 accumulation, ONE device->host copy at the end instead of
one 4.7/75 MB copy per hook call) and the driver is a loop over synthetic COCO-shaped batches (the data modules are
outside the hot path).  Output: `<log_dir>/<representation_name>.pth` = torch.save(dict name -> float64 [D,D]),
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


def main(argv):
    ge.import_package()
    cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
    vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
    vu = importlib.import_module("vl_merging_amd.vilt.modules.vilt_utils")
    from bench import synthetic_batch
    batches = 4
    rest = []
    for a in argv:
        if a.startswith("batches="):
            batches = int(a.split("=", 1)[1])
        else:
            rest.append(a)
    cfg = cfgmod.parse_cli(rest)
    torch.cuda.set_device(0)
    torch.manual_seed(cfg["seed"])
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).cuda().eval()
    model.setup_engine()
    vu.set_task(model)
    cap = model.start_gram_capture()
    B = cfg["per_gpu_batchsize"] or 2
    with torch.no_grad():
        for i in range(batches):
            batch = synthetic_batch(B, cfg["image_size"], cfg["max_text_len"], cfg["vocab_size"], 4321 + i, "cuda")["vl"]
            model(batch)
    model.stop_gram_capture()
    os.makedirs(cfg["log_dir"], exist_ok=True)
    path = os.path.join(cfg["log_dir"], cfg["representation_name"] + ".pth")
    grams = cap.state_dict()
    torch.save(grams, path)
    for k, v in list(grams.items())[:4]:
        print(k, tuple(v.shape), float(v.min()), float(v.max()))
    print("saved %d gram matrices to %s" % (len(grams), path))


if __name__ == "__main__":
    main(sys.argv[1:])
