"""A few GEMM launches for PMC collection (rocprofv3 --pmc ...): qkv / fc1 / fc2 forward shapes of the fused pass."""
import importlib, sys, os, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.import_package()
ops = importlib.import_module("vl_merging_amd.ops")
M = int(os.environ.get("PMC_M", 88 * 617))
for name, ta, tb, m, n, k in [("qkv", False, False, M, 2304, 768), ("fc1", False, False, M, 3072, 768),
                               ("fc2fwd", False, False, M, 768, 3072), ("fc2dgrad", False, True, M, 3072, 768)]:
    a = torch.randn((k, m) if ta else (m, k), device="cuda").to(torch.bfloat16)
    b = torch.randn((k, n) if tb else (n, k), device="cuda").to(torch.bfloat16)
    out = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, out, ta, tb)
torch.cuda.synchronize()
