"""In-kernel timeline of the GEMM tile (diagnostic build, never the product): prologue / main loop / epilogue cycles
per workgroup from s_memtime stamps.  Build (CPU container):  python tools/stamp_gemm.py build
Run (GPU box):                                               python tools/stamp_gemm.py [M]"""
import ctypes
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vl-merging_amd")
LIB = os.path.join(PKG, "lib", "libvlm_hip_stamps.so")


VARIANTS = {"": []}  # (the knock-out variants of rounds 1-3 are gone from the sources: DESIGN.md 4.1)


def build():
    sys.path.insert(0, PKG)
    import build_ext as B
    B.build(verbose=False)
    for suffix, flags in VARIANTS.items():
        obj = os.path.join(B.BUILD, "gemm_stamps%s.o" % suffix)
        subprocess.run([B.HIPCC] + B.COMMON + ["-DVLM_DIAG"] + flags + ["-c", os.path.join(B.CSRC, "gemm.hip"), "-o", obj],
                       check=True, capture_output=True)
        objs = [os.path.join(B.BUILD, f) for f in os.listdir(B.BUILD) if f.endswith(".hip.o") and f != "gemm.hip.o"] + [obj]
        lib = LIB.replace(".so", suffix + ".so")
        subprocess.run([B.HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs, check=True)
        print("built", lib)


def main():
    os.environ["VLM_LIB_PATH"] = LIB.replace(".so", os.environ.get("STAMP_VARIANT", "") + ".so")
    print("variant:", os.environ.get("STAMP_VARIANT", "") or "full")
    import torch
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    ge.import_package()
    ops = importlib.import_module("vl_merging_amd.ops")
    L = importlib.import_module("vl_merging_amd._lib")
    lib = L.get_lib()
    lib.vlm_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    lib.vlm_debug_set_stamp_buffer.restype = None
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 88 * 617
    bf = torch.bfloat16
    dev = "cuda"

    def t(shape, dt=bf):
        return torch.randn(*shape, device=dev).to(dt)

    x768, x3072 = t((M, 768)), t((M, 3072))
    w_qkv, w_fc1, w_fc2 = t((2304, 768)), t((3072, 768)), t((768, 3072))
    res = t((M, 768), torch.float32)
    b768, b3072, gam = t((768,), torch.float32), t((3072,), torch.float32), t((768,), torch.float32)
    o2304, o3072, aux = torch.empty(M, 2304, device=dev, dtype=bf), torch.empty(M, 3072, device=dev, dtype=bf), t((M, 3072))
    cases = [
        ("qkv fwd (bias)", lambda: ops.gemm(x768, w_qkv, o2304, bias=b768.repeat(3)), 2304),
        ("fc1 fwd bias+gelu+aux", lambda: ops.gemm(x768, w_fc1, o3072, bias=b3072, act=L.ACT_GELU, aux=aux), 3072),
        ("fc2 fwd f32+res", lambda: ops.gemm(x3072, w_fc2, res, bias=b768, col_scale=gam, residual=res), 768),
        ("fc2 dgrad gelu_bwd", lambda: ops.gemm(x768, w_fc2, o3072, False, True, act=L.ACT_GELU_BWD, aux=aux), 3072),
        ("fc1 dgrad", lambda: ops.gemm(x3072, w_fc1, torch.empty(M, 768, device=dev, dtype=bf), False, True), 768),
    ]
    for name, fn, n in cases:
        tm_, tn_ = int(os.environ.get("STAMP_TILE_M", 128)), int(os.environ.get("STAMP_TILE_N", 128))
        nwg = ((M + tm_ - 1) // tm_) * ((n + tn_ - 1) // tn_)
        if os.environ.get("STAMP_PERSISTENT"):
            nwg = min(nwg, 256)
        st = torch.zeros(nwg * 8, device=dev, dtype=torch.int64)
        for _ in range(3):
            fn()
        lib.vlm_debug_set_stamp_buffer(st.data_ptr())
        fn()
        torch.cuda.synchronize()
        lib.vlm_debug_set_stamp_buffer(None)
        s = st.view(nwg, 8).cpu().double()
        pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
        tot = s[:, 3] - s[:, 0]
        real = (s[:, 5] - s[:, 4]) * 10.0  # ns (100 MHz)
        if os.environ.get("STAMP_PERSISTENT"):
            print("%-24s persistent: loop %6.0f cyc/tile, epilogue(issue) %6.0f cyc/tile, tiles/WG %.1f, WG total %.0f cyc, clock %.2f GHz, span %.1f us"
                  % (name, loop.median(), epi.median(), s[:, 7].median(), s[:, 6].median(),
                     (s[:, 6] / real).median(), (s[:, 5].max() - s[:, 4].min()) * 10.0 / 1e3))
            continue
        clk = (tot / real).median()
        span_ns = (s[:, 5].max() - s[:, 4].min()) * 10.0
        print("%-24s wgs %5d  prologue %6.0f  loop %6.0f  epilogue %6.0f  total %6.0f cyc (medians)  clock %.2f GHz  "
              "kernel span %.1f us  sum(tot)/span/CU = %.2f WG resident" %
              (name, nwg, pro.median(), loop.median(), epi.median(), tot.median(), clk, span_ns / 1e3,
               real.sum() / span_ns / 256))
        # first-wave (cold) vs steady state
        order = s[:, 4].argsort()
        late = order[nwg // 2:]
        print("   steady-state half: prologue %6.0f loop %6.0f epilogue %6.0f" %
              (pro[late].median(), loop[late].median(), epi[late].median()))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build()
    else:
        main()
