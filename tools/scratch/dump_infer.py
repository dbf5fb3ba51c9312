"""Dump model.infer(...) text/image features of the full-size eval pass (B = 22, all_moe or ufo) to a file: tools/scratch/dump_infer.py OUT [arch] [B]"""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
ge.import_package()
import test_model_gpu as T
mods = (importlib.import_module("vl_merging_amd.vilt.config"), importlib.import_module("vl_merging_amd.vilt.modules.vilt_module"))
arch = sys.argv[2] if len(sys.argv) > 2 else "all_moe"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 22
torch.manual_seed(11)
model = T.build_base(mods, arch, T.GOLDEN if hasattr(T, "GOLDEN") else os.path.join("/root/repo/tests", "golden"), {"itm": 1, "mlm": 1, "ifm": 1}, tag=None, max_vl=40, train=False)
torch.manual_seed(3)
nb = T.det_batch(B, 384, 40, 1024, seed=2024 + B)
batch = T.gpu_batch(nb)
with torch.no_grad():
    got = model.infer(batch)
torch.cuda.synchronize()
np.savez(sys.argv[1], text=got["text_feats"].float().cpu().numpy(), image=got["image_feats"].float().cpu().numpy())
