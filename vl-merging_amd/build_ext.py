"""Build libvlm_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; objects are cached by mtime under csrc/_build/.
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "lib", "libvlm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-ffp-contract=off"]  # no implicit FMA anywhere: kernels that want FMA call fmaf/MFMA explicitly


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "vlm_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src):
    obj = os.path.join(BUILD, os.path.basename(src) + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _deps_mtime()):
        return obj, False
    cmd = [HIPCC] + COMMON + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout[-2000:], r.stderr[:3000]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr[-4000:])
    return obj, True


def build(verbose=True, jobs=None):
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with cf.ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    rebuilt = any(c for _, c in res)
    if rebuilt or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-8000:])
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build()
