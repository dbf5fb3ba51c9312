"""Which hipBLASLt solution (macro tile, grid) torch.matmul picks for the block GEMM shapes.  Run under
rocprofv3 --kernel-trace --output-format csv; tools/vendor_gemm_names.sh prints name / grid / workgroup per shape."""
import sys
import torch
shapes = [("fc2 fwd", 54296, 768, 3072), ("fc1 dgrad", 54296, 768, 3072), ("qkv dgrad", 54296, 768, 2304), ("proj fwd", 54296, 768, 768),
          ("fc2 fwd s", 13574, 768, 3072), ("proj fwd s", 13574, 768, 768)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        c = torch.matmul(a, b.t())
    torch.cuda.synchronize()
