"""Row-kernel micro-benchmark (LayerNorm fwd/bwd, LayerScale bwd, colsum) at the fused-pass size."""
import importlib, sys, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.import_package()
ops = importlib.import_module("vl_merging_amd.ops")

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M in (22 * 617, 88 * 617):
    D = 768
    x = torch.randn(M, D, device="cuda"); g = torch.ones(D, device="cuda"); b = torch.zeros(D, device="cuda")
    y = torch.empty(M, D, device="cuda", dtype=torch.bfloat16); st = torch.empty(M, 2, device="cuda")
    dy = torch.randn(M, D, device="cuda").to(torch.bfloat16); dres = torch.randn(M, D, device="cuda")
    dx = torch.empty(M, D, device="cuda"); dg = torch.zeros(D, device="cuda"); db = torch.zeros(D, device="cuda")
    dh = torch.randn(M, 3072, device="cuda").to(torch.bfloat16); cs = torch.zeros(3072, device="cuda")
    rs = torch.ones(M, device="cuda")
    t = timeit(lambda: ops.layernorm_fwd(x, g, b, 1e-6, y, st)); print("M=%d ln_fwd  %.1f us  %.2f TB/s" % (M, t, M * D * 6 / t / 1e6))
    t = timeit(lambda: ops.layernorm_bwd(dy, x, st, g, dx, dres=dres, dgamma=dg, dbeta=db)); print("M=%d ln_bwd  %.1f us  %.2f TB/s" % (M, t, M * D * 14 / t / 1e6))
    t = timeit(lambda: ops.layerscale_bwd(dx, y, g, rs, dy, dg, db)); print("M=%d scale_bwd %.1f us  %.2f TB/s" % (M, t, M * D * 8 / t / 1e6))
    sdy = torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: ops.layernorm_bwd_scale(dy, x, st, g, dx, dres, dg, db, y=y, sgamma=g, row_scale=rs, sdy=sdy, dsgamma=dg, dsbias=db))
    print("M=%d ln_bwd + scale_bwd in one pass %.1f us  %.2f TB/s" % (M, t, M * D * 18 / t / 1e6))
    t = timeit(lambda: ops.colsum(dh, cs)); print("M=%d colsum3072 %.1f us  %.2f TB/s" % (M, t, M * 3072 * 2 / t / 1e6))
