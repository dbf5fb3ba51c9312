"""Retrieval-evaluation throughput (compute_irtr_recall): base_vl ufo at 384^2, synthetic COCO-1k-shaped
set (1000 images x 5 captions by default), batches of 32 as in the reference (objectives.py:587,613)."""
import importlib
import json
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge

ge.import_package()
cfgmod = importlib.import_module("vl_merging_amd.vilt.config")
vm = importlib.import_module("vl_merging_amd.vilt.modules.vilt_module")
obj = importlib.import_module("vl_merging_amd.vilt.modules.objectives")


def main():
    n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    caps, bs = 5, 32
    cfg = cfgmod.make_config("task_finetune_irtr_coco_square_randaug_base_image384", "ufo", image_size=384, vit="vit_base_patch16_384")
    torch.manual_seed(0)
    model = vm.ViLTransformerSS(cfg, *cfgmod.routing_configs(cfg)).cuda().eval()
    model.setup_engine()
    g = torch.Generator().manual_seed(1)
    T = cfg["max_text_len"]
    texts, images = [], []
    for lo in range(0, n_img * caps, bs):
        n = min(bs, n_img * caps - lo)
        ids = torch.randint(1000, cfg["vocab_size"], (n, T), generator=g)
        ids[:, 0] = 101
        texts.append({"text_ids": ids.cuda(), "text_masks": torch.ones(n, T, dtype=torch.long).cuda(),
                      "text_labels": torch.full((n, T), -100).cuda(), "img_index": [(lo + j) // caps for j in range(n)]})
    for lo in range(0, n_img, bs):
        n = min(bs, n_img - lo)
        images.append({"image": [(torch.rand(n, 3, 384, 384, generator=g) * 2 - 1).cuda()], "img_index": list(range(lo, lo + n)),
                       "text_masks": torch.ones(1, T, dtype=torch.long).cuda()})
    obj.compute_irtr_recall(model, texts[:2], images[:2] if n_img >= 64 else images)  # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = obj.compute_irtr_recall(model, texts, images)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": "compute_irtr_recall base_vl ufo 384^2, %d images x %d captions, batch 32" % (n_img, caps),
                      "seconds": dt, "images_per_s_whole_eval": n_img / dt, "recalls": [float(x) for x in out[:6]]}))


if __name__ == "__main__":
    main()
