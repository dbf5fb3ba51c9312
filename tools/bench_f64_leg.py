"""The fp64 leg of BASELINE configs[3] on its own (for a rocprofv3 kernel trace): RegMean at base size with device-resident inputs
(vilt_module.py:366-531) twice, and the Gram capture SYRK (cache_gram_matrices.py:246-254) on one hooked input of the 88-sample
pass at D = 768 and D = 3072, four times each.  Prints seconds and fp64 TFLOP/s against the 78.6 TFLOP/s MFMA roofline."""
import importlib
import sys
import time

import torch

sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge

ge.import_package()
bm = importlib.import_module("vl_merging_amd.bench_merge")
rg = importlib.import_module("vl_merging_amd.regmean")
ops = importlib.import_module("vl_merging_amd.ops")
PEAK = 78.6


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
    cfg = dict(vlffn_start_layer_index=10, only_activate_used_experts=False, scaling_for_non_diag=0.9, loss_names={"irtr": 1},
               gram_matrices=None)
    rg.regmean(sd, cfg, gram_matrices=grams)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = rg.regmean(sd, cfg, gram_matrices=grams)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fl = sum(2 * 2.0 * v.shape[0] * v.shape[1] ** 2 + v.shape[1] ** 3 / 3.0 + 2.0 * v.shape[0] * v.shape[1] ** 2
             for k, v in out.items() if k.endswith(".weight") and v.dim() == 2 and "norm" not in k and "blocks" in k)
    print("regmean base size: %.3f s, %.2f fp64 TFLOP/s = %.3f of %.1f" % (dt, fl / dt / 1e12, fl / dt / 1e12 / PEAK, PEAK))
    for D in (768, 3072):
        x = torch.randn(54296, D, device="cuda").to(torch.bfloat16)
        g = torch.zeros(D, D, device="cuda", dtype=torch.float64)
        ops.gram_accumulate(x, g)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            ops.gram_accumulate(x, g)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        fl = 2.0 * 54296 * D * D
        print("gram capture D=%d, 54296 rows: %.3f ms, %.1f TFLOP/s over 2 M D^2 = %.3f of %.1f" % (D, dt * 1e3, fl / dt / 1e12, fl / dt / 1e12 / PEAK, PEAK))


if __name__ == "__main__":
    main()
