"""Persistent 256x256 kernel: results with a small CU budget (several tiles per workgroup) against the one-tile-per-workgroup
launch of the same library (budget = all CUs, tiles <= CUs) and against an fp32 torch reference, every epilogue variant."""
import importlib, sys
import torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.import_package()
ops = importlib.import_module("vl_merging_amd.ops")
L = importlib.import_module("vl_merging_amd._lib")
lib = L.get_lib()
torch.manual_seed(0)
dev = "cuda"
bf = torch.bfloat16
lib.vlm_gemm_set_big_tile_mode(2)
bad = 0
for (M, N, K) in ((1500, 768, 768), (2000, 1024, 256), (3100, 512, 1024), (700, 256, 128)):
    a = torch.randn(M, K, device=dev).to(bf); w = (torch.randn(N, K, device=dev) * 0.05).to(bf)
    bias = torch.randn(N, device=dev); gam = torch.randn(N, device=dev); rs = torch.rand(M, device=dev)
    res = torch.randn(M, N, device=dev); aux_in = torch.randn(M, N, device=dev).to(bf)
    def run(variant):
        if variant == 0:
            o = torch.empty(M, N, device=dev, dtype=bf); ops.gemm(a, w, o, bias=bias); return (o,)
        if variant == 1:
            o = torch.empty(M, N, device=dev, dtype=bf); h = torch.empty(M, N, device=dev, dtype=bf)
            ops.gemm(a, w, o, bias=bias, act=L.ACT_GELU_DERIV, aux=h); return (o, h)
        if variant == 2:
            o = torch.empty(M, N, device=dev, dtype=bf); cs = torch.zeros(N, device=dev)
            ops.gemm(a, w, o, act=L.ACT_MUL_AUX, aux=aux_in, col_sum=cs); return (o, cs)
        if variant == 3:
            o = torch.empty(M, N, device=dev); y = torch.empty(M, N, device=dev, dtype=bf)
            ops.gemm(a, w, o, bias=bias, col_scale=gam, row_scale=rs, residual=res, aux=y); return (o, y)
        if variant == 4:
            o = torch.empty(M, N, device=dev); ops.gemm(a, w, o, bias=bias, col_scale=gam, residual=res); return (o,)
        o = torch.empty(M, N, device=dev); ops.gemm(a, w, o, bias=bias); return (o,)
    for variant in range(6):
        lib.vlm_set_cu_budget(0)
        ref = run(variant)
        for cus in (1, 3, 8):
            lib.vlm_set_cu_budget(cus)
            got = run(variant)
            torch.cuda.synchronize()
            for r, g in zip(ref, got):
                if variant == 2 and r.dim() == 1:
                    ok = torch.allclose(r, g, rtol=1e-4, atol=1e-2)
                else:
                    ok = torch.equal(r, g)
                if not ok:
                    bad += 1
                    print("MISMATCH", (M, N, K), "variant", variant, "cus", cus, (r.float() - g.float()).abs().max().item())
    lib.vlm_set_cu_budget(0)
    o = torch.empty(M, N, device=dev, dtype=bf); ops.gemm(a, w, o, bias=bias)
    want = a.float() @ w.float().t() + bias
    err = (o.float() - want).abs().max().item()
    print((M, N, K), "max abs err vs fp32 torch", err)
    if err > 0.1: bad += 1
print("persist_check:", "FAILED %d" % bad if bad else "ok")
sys.exit(1 if bad else 0)
