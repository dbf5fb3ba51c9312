"""K-split tail of the 256x256 kernel: the fp32 residual-stream GEMM (fc2 forward, folded LayerScale) split vs whole tiles."""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.import_package()
ops = importlib.import_module("vl_merging_amd.ops")
L = importlib.import_module("vl_merging_amd._lib")
lib = L.get_lib()

def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N, K in ((54296, 768, 3072), (54296, 768, 2304), (54296, 768, 768), (13574, 768, 3072)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    x = torch.randn(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    rs = torch.ones(M, device="cuda")
    r = {}
    for mode in (1, 2, 1, 2):
        lib.vlm_gemm_set_big_tile_mode(mode)
        t = timeit(lambda: ops.gemm(a, b, x, bias=bias, row_scale=rs, residual=x))
        r.setdefault(mode, []).append(t)
    lib.vlm_gemm_set_big_tile_mode(-1)
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d  fp32 residual epilogue: default (split where planned) %s us, whole tiles %s us  -> %.0f / %.0f TFLOP/s"
          % (M, N, K, ["%.1f" % v for v in r[1]], ["%.1f" % v for v in r[2]], fl / min(r[1]) / 1e6, fl / min(r[2]) / 1e6))
