"""Build libvlm_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; objects are cached by mtime under csrc/_build/.
"""
import concurrent.futures as cf
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "lib", "libvlm_hip.so")
RESOURCES = os.path.join(HERE, "lib", "kernel_resources.json")  # per-kernel register / LDS use of the last full compile
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-ffp-contract=off"]  # no implicit FMA anywhere: kernels that want FMA call fmaf/MFMA explicitly


def _deps_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]  # .inc: the generated asm streams
    hs.append(os.path.join(HERE, "..", "include", "vlm_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src):
    obj = os.path.join(BUILD, os.path.basename(src) + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), _deps_mtime()):
        return obj, False
    cmd = [HIPCC] + COMMON + ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout[-2000:], r.stderr[:3000]))
    res, name, rest, rest_skip = {}, None, [], False
    for line in r.stderr.splitlines():  # the register allocator's verdict per kernel (tests/test_build_cpu.py watches it)
        if "-Rpass-analysis=kernel-resource-usage" not in line:
            if not (rest_skip and re.match(r"^\s*(\d+\s*)?\|", line)):  # the source excerpt under a remark
                rest.append(line)
            continue
        rest_skip = True
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            res[name] = {}
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and name:
            res[name][m.group(1).strip()] = int(m.group(2))
    with open(obj + ".resources.json", "w") as f:
        json.dump(res, f)
    if "\n".join(rest).strip():
        sys.stderr.write("\n".join(rest)[-4000:] + "\n")
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
    merged = {}
    for o in objs:
        try:
            with open(o + ".resources.json") as f:
                merged.update(json.load(f))
        except OSError:
            pass
    if len(merged) and all(os.path.exists(o + ".resources.json") for o in objs):
        # rewritten on EVERY build (its mtime tells tests/test_build_cpu.py the record is not older than the sources)
        with open(RESOURCES, "w") as f:
            json.dump(merged, f, indent=0, sort_keys=True)
    return LIB


if __name__ == "__main__":
    build()
