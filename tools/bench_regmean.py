"""RegMean merge at base size with device-resident inputs (the reference takes 48 s on the CPU):
(sum_i W_i G_i)(sum_i G_i)^-1 in fp64 for the 48 weight matrices + HIP averages for the rest."""
import importlib
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge

ge.import_package()
bm = importlib.import_module("vl_merging_amd.bench_merge")
rg = importlib.import_module("vl_merging_amd.regmean")


def main():
    sd = bm.synthetic_all_moe_blocks()
    gen = torch.Generator(device="cuda").manual_seed(3)
    grams = {}
    for k, v in sd.items():
        if k.endswith(".weight") and "norm" not in k and ".vl." not in k:
            name = k.replace(".qkv.weight", "") if "qkv" in k else k.replace(".weight", "")
            D = v.shape[1]
            x = torch.randn(D + 64, D, device="cuda", dtype=torch.float64, generator=gen)
            grams[name] = x.t() @ x
    cfg = dict(vlffn_start_layer_index=10, only_activate_used_experts=False, scaling_for_non_diag=0.9,
               loss_names={"irtr": 1}, gram_matrices=None)
    rg.regmean(sd, cfg, gram_matrices=grams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = rg.regmean(sd, cfg, gram_matrices=grams)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    flops = sum(2.0 * v.shape[0] * v.shape[1] ** 2 * 3 + (2.0 / 3) * v.shape[1] ** 3 for k, v in out.items()
                if k.endswith(".weight") and v.dim() == 2 and "norm" not in k and "blocks" in k)
    print("regmean base size, fp64, device-resident inputs: %.3f s  (~%.1f fp64 TFLOP/s over W.G, inverse and product)" %
          (dt, flops / dt / 1e12))


if __name__ == "__main__":
    main()
