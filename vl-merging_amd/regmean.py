"""RegMean merge (reference src/vilt/modules/vilt_module.py:366-531) on the GPU.

Biases / q_bias / v_bias / LayerNorm tensors: plain averages through the HIP merge kernel (VLM_MERGE_MEAN,
bit-exact with the reference: add in order, then divide by the count).  Linear weights:
W* = (sum_m W_m G'_m)(sum_m G'_m)^-1 with G' = a*G + (1-a)*diag(G), kept in float64 like the reference.
Round 1: the fp64 products and the inverse run through torch's ROCm libraries on the device (rocBLAS / rocSOLVER);
a hand-written v_mfma_f64 SYRK/GEMM + Cholesky path is the next step for this row (DESIGN.md "RegMean").
"""
import torch

from . import _lib as L
from . import merge as M


def scale_gram(G, alpha):
    """:388-392"""
    return alpha * G + (1 - alpha) * torch.diag_embed(torch.diag(G))


def regmean(state_dict, config, gram_matrices=None, device="cuda", plan_out=None):
    out = M._passthrough(state_dict)
    if gram_matrices is None:
        from . import checkpoint
        gram_matrices = checkpoint.load_file(config["gram_matrices"])
    alpha = config["scaling_for_non_diag"]
    plan = M.MergePlan(device)
    dev = plan.device
    for i in range(M.NUM_MERGE_LAYERS):
        mods = M.modalities_for_layer(config, i, honour_only_used=False)
        for src, dst in M._tensor_names(i):
            is_weight = dst.endswith(".weight") and "norm" not in dst
            if not is_weight:
                srcs, through = M._collect(state_dict, src, dst, mods)
                out[dst] = through if srcs is None else plan.add(L.MERGE_MEAN, [t for _, t in srcs], None)
                continue
            num, den, through = 0, 0, None
            for m in mods:
                name = src(m)
                gname = name.replace(".qkv.weight", "") if "qkv" in name else name.replace(".weight", "")
                if name in state_dict:
                    if gname not in gram_matrices:
                        continue  # :419-420 (vl experts never get a gram)
                    G = scale_gram(gram_matrices[gname].to(dev, torch.float64), alpha)
                    den = den + G
                    num = num + state_dict[name].to(dev, torch.float64) @ G
                else:
                    through = state_dict[dst]
                    break
            if through is not None:
                out[dst] = through
            elif isinstance(den, int):
                out[dst] = num
            else:
                out[dst] = num @ torch.inverse(den)  # stays float64 in the returned dict (:432-434)
    if plan.jobs:
        plan.run()
    if plan_out is not None:
        plan_out.append(plan)
    return out
