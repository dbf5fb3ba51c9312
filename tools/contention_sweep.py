"""Single-GPU stand-in for RCCL's contention during backward (VERDICT r4 #3b): bench.py's training step with every gradient
bucket's collective replaced by k workgroups that hold a CU each for the bucket's transfer time plus a device copy of the
bucket on the communication stream (VLM_DDP_STANDIN=k, ddp.FlatGradReducer(standin=...)), against the two switches a
data-parallel rank has: VLM_WGRAD_STREAM (weight gradients on a side stream) and VLM_GEMM_CUS (CUs the GEMM grids plan for).
Writes gpurun_out/r05/contention.json (copied to profiles/r05_contention.json).   python tools/contention_sweep.py [--task irtr]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
extra = sys.argv[1:]
out = {"command": "bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer " + " ".join(extra),
       "standin": "k workgroups x 512 threads x 64 KB LDS spinning for 2*(7/8)*bucket bytes / 300 GB/s + one copy of the bucket, per bucket",
       "runs": []}
combos = [(0, ws, 256) for ws in (1, 0)]
for k in (8, 16, 32):
    for ws in (1, 0):
        for cus in (256, 248, 240, 224):
            combos.append((k, ws, cus))
for k, ws, cus in combos:
    env = dict(os.environ, VLM_WGRAD_STREAM=str(ws), VLM_GEMM_CUS=str(cus))
    if k:
        env["VLM_DDP_STANDIN"] = str(k)
    else:
        env.pop("VLM_DDP_STANDIN", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--no-merge",
                        "--no-calibrate", "--no-secondary", "--no-gemm-timer"] + extra, capture_output=True, text=True, env=env, cwd=ROOT)
    rec = {"standin_cus": k, "wgrad_stream": ws, "gemm_cus": cus}
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        rec.update(ms_per_step=d["ms_per_step"], samples_per_s=d["value"], exposed_comm_ms=d["exposed_comm_ms_per_step"])
    except Exception as e:
        rec["error"] = repr(e) + r.stderr[-300:]
    out["runs"].append(rec)
    print(rec, flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r05"), exist_ok=True)
name = "contention%s.json" % ("_" + "_".join(a.strip("-") for a in extra) if extra else "")
with open(os.path.join(ROOT, "gpurun_out", "r05", name), "w") as f:
    json.dump(out, f, indent=1)
