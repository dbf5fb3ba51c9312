"""Merge-kernel measurement (BASELINE config 4): base-size all_moe -> ufo interpolation, inputs resident in HBM."""
import statistics

import torch

from . import merge as M

D, F = 768, 3072
ALGO_BYTES = 1077239808  # 737 058 816 read + 340 180 992 written


def _expert_block(i, m, gen):
    pre = f"transformer.blocks.{i}."
    mm = m + "."
    shapes = {
        pre + f"attn.{mm}q_bias": (D,), pre + f"attn.{mm}v_bias": (D,), pre + f"attn.{mm}qkv.weight": (3 * D, D),
        pre + f"attn.{mm}proj.weight": (D, D), pre + f"attn.{mm}proj.bias": (D,),
        pre + f"norm1.{mm}weight": (D,), pre + f"norm1.{mm}bias": (D,),
        pre + f"mlp.{mm}fc1.weight": (F, D), pre + f"mlp.{mm}fc1.bias": (F,),
        pre + f"mlp.{mm}fc2.weight": (D, F), pre + f"mlp.{mm}fc2.bias": (D,),
        pre + f"norm2.{mm}weight": (D,), pre + f"norm2.{mm}bias": (D,),
    }
    return {k: torch.randn(s, device="cuda", generator=gen) * 0.02 for k, s in shapes.items()}


def synthetic_all_moe_blocks(seed=0):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    sd = {}
    for i in range(12):
        for m in ["v", "l"] + (["vl"] if i >= 10 else []):
            sd.update(_expert_block(i, m, gen))
    return sd


def run(reps=20, warmup=3, ratio=0.5, check_layers=None):
    """`check_layers`: also return, under "check", (the inputs of those layers, the merged tensors of those layers as they
    stand in the output buffers AFTER the last timed launch, the merge config) so that the caller can compare what was timed
    with a checker of its own (bench.py: the CPU oracle; this package never imports one)."""
    sd = synthetic_all_moe_blocks()
    cfg = dict(vlffn_start_layer_index=10, only_activate_used_experts=False, merge_ratio=ratio, loss_names={})
    plans = []
    merged = M.merge_weights(sd, cfg, plan_out=plans)
    plan = plans[0]
    assert plan.bytes_read + plan.bytes_written == ALGO_BYTES
    for _ in range(warmup):
        plan.run()
    torch.cuda.synchronize()
    times = []
    for _ in range(reps):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.run()
        e1.record()
        e1.synchronize()
        times.append(e0.elapsed_time(e1) * 1e-3)
    med = statistics.median(times)
    res = {"kernel": "vlm_merge_kernel", "seconds_median": med, "seconds_min": min(times),
           "algorithmic_bytes": ALGO_BYTES, "GBps": ALGO_BYTES / med / 1e9, "reps": reps}
    if check_layers is not None:
        def layer(k):
            return int(k.split(".")[2])
        res["check"] = ({k: v for k, v in sd.items() if layer(k) in check_layers},
                        {k: v for k, v in merged.items() if "transformer.blocks." in k and layer(k) in check_layers}, cfg,
                        tuple(check_layers))
    return res


if __name__ == "__main__":
    print(run())
