"""GEMM epilogue micro-benchmark: the in-situ epilogue configurations of the fused 4B-sample pass (M = 88*617)."""
import importlib
import sys

import torch

sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge

ge.import_package()
ops = importlib.import_module("vl_merging_amd.ops")
L = importlib.import_module("vl_merging_amd._lib")
from tools.bench_gemm import timeit  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 88 * 617
    dev = "cuda"
    bf = torch.bfloat16

    def t(shape, dt=bf):
        return torch.randn(*shape, device=dev).to(dt)

    x768, x3072 = t((M, 768)), t((M, 3072))
    w_proj, w_fc1, w_fc2 = t((768, 768)), t((3072, 768)), t((768, 3072))
    res = t((M, 768), torch.float32)
    bias768, bias3072, gam = t((768,), torch.float32), t((3072,), torch.float32), t((768,), torch.float32)
    o768f, o768b, o3072b, aux = torch.empty(M, 768, device=dev), torch.empty(M, 768, device=dev, dtype=bf), \
        torch.empty(M, 3072, device=dev, dtype=bf), t((M, 3072))
    cases = [
        ("proj plain bf16", lambda: ops.gemm(x768, w_proj, o768b), 768, 768),
        ("proj plain f32", lambda: ops.gemm(x768, w_proj, o768f), 768, 768),
        ("proj f32 +res(in place)", lambda: ops.gemm(x768, w_proj, res, bias=bias768, col_scale=gam, residual=res), 768, 768),
        ("proj f32 +res(out of place)", lambda: ops.gemm(x768, w_proj, o768f, bias=bias768, col_scale=gam, residual=res), 768, 768),
        ("fc2 plain bf16", lambda: ops.gemm(x3072, w_fc2, o768b), 768, 3072),
        ("fc2 f32 +res(in place)", lambda: ops.gemm(x3072, w_fc2, res, bias=bias768, col_scale=gam, residual=res), 768, 3072),
        ("fc1 plain bf16", lambda: ops.gemm(x768, w_fc1, o3072b), 3072, 768),
        ("fc1 bias+gelu", lambda: ops.gemm(x768, w_fc1, o3072b, bias=bias3072, act=L.ACT_GELU), 3072, 768),
        ("fc1 bias+gelu+aux", lambda: ops.gemm(x768, w_fc1, o3072b, bias=bias3072, act=L.ACT_GELU, aux=aux), 3072, 768),
        ("fc2 dgrad plain", lambda: ops.gemm(x768, w_fc2, o3072b, False, True), 3072, 768),
        ("fc2 dgrad gelu_bwd", lambda: ops.gemm(x768, w_fc2, o3072b, False, True, act=L.ACT_GELU_BWD, aux=aux), 3072, 768),
        ("fc1 bias+gelu+deriv", lambda: ops.gemm(x768, w_fc1, o3072b, bias=bias3072, act=L.ACT_GELU_DERIV, aux=aux), 3072, 768),
        ("fc2 dgrad mul_aux", lambda: ops.gemm(x768, w_fc2, o3072b, False, True, act=L.ACT_MUL_AUX, aux=aux), 3072, 768),
    ]
    for name, fn, n, k in cases:
        us = timeit(fn)
        print("%-28s N=%5d K=%5d %8.1f us  %7.1f TFLOP/s" % (name, n, k, us, 2.0 * M * n * k / us / 1e6))


if __name__ == "__main__":
    main()
