"""GEMM micro-benchmark at the training shapes (M = 22*617 tokens, or argv[1]): vlm_gemm_bf16 next to the vendor library
(hipBLASLt through torch.matmul, same operand layouts, bf16 output) on the same box, shape by shape."""
import importlib
import sys

import torch

sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge

ge.import_package()
ops = importlib.import_module("vl_merging_amd.ops")
L = importlib.import_module("vl_merging_amd._lib")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 22 * 617
    shapes = [("qkv fwd", False, False, M, 2304, 768), ("proj fwd", False, False, M, 768, 768),
              ("fc1 fwd", False, False, M, 3072, 768), ("fc2 fwd", False, False, M, 768, 3072),
              # dgrad as the engine runs it: against the bf16 TRANSPOSED weight shadow (engine.wT16), i.e. both operands
              # K-contiguous like a forward call (the K-strided-B form these lines used until round 4 is a path no training
              # step takes: 392 instead of 256 us for fc2 dgrad at M = 54 296)
              ("fc1 dgrad", False, False, M, 768, 3072), ("fc2 dgrad", False, False, M, 3072, 768),
              ("qkv dgrad", False, False, M, 768, 2304), ("proj dgrad", False, False, M, 768, 768),
              ("fc1 wgrad", True, True, 3072, 768, M), ("fc2 wgrad", True, True, 768, 3072, M),
              ("qkv wgrad", True, True, 2304, 768, M), ("square 4096", False, False, 4096, 4096, 4096),
              ("text qkv", False, False, 880, 2304, 768)]
    tot_t = tot_f = tot_v = 0
    for name, ta, tb, m, n, k in shapes:
        a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
        b = torch.randn((k, n) if tb else (n, k), device="cuda").to(torch.bfloat16)
        out = torch.empty(m, n, device="cuda", dtype=torch.float32 if ta else torch.bfloat16)
        t = timeit(lambda: ops.gemm(a, b, out, ta, tb, accumulate=bool(ta)))
        fl = 2.0 * m * n * k
        am = a.t() if ta else a          # [m, k] view in the stored layout
        bm = b if tb else b.t()          # [k, n] view
        ref = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
        tv = timeit(lambda: torch.matmul(am, bm, out=ref))
        print("%-12s M=%6d N=%5d K=%6d  %8.1f us  %7.1f TFLOP/s   | hipBLASLt %8.1f us  %7.1f TFLOP/s  (ours / vendor = %.2f)"
              % (name, m, n, k, t, fl / t / 1e6, tv, fl / tv / 1e6, tv / t))
        if "square" not in name:
            tot_t += t; tot_f += fl; tot_v += tv
    print("weighted over training shapes: %.1f TFLOP/s (hipBLASLt: %.1f TFLOP/s)" % (tot_f / tot_t / 1e6, tot_f / tot_v / 1e6))


if __name__ == "__main__":
    main()
