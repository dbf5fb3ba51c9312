"""bench.py contract (the driver parses its LAST stdout line): one JSON object with the agreed fields, a roofline and a
merge object; run at 2 timed steps so the test stays under a minute."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--no-calibrate", "--gemm-timer-every", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = r.stdout.strip().splitlines()[-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "merge"):
        assert k in d, k
    assert d["unit"] == "samples/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["dtype"] == "bf16" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 50 and abs(d["value"] - 22 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"] + 1e-3
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.05 < rf["frac"] < 1.0
    # the committed PMC file vouches for this build of the kernels (a stale one is refused: traffic None, "stale" says why)
    src = rf["traffic_source"]
    assert src["file"].startswith("profiles/r") and len(src["git_blob"]) == 40 and "stale" not in src, src
    assert rf["traffic"] is not None and rf["traffic"] > 0
    # the attention kernels north_star puts a number on, HIP-event-bracketed in this run
    ra = d["roofline_attention"]
    for kind in ("fwd", "bwd"):
        a = ra[kind]
        assert a["bound"] == "mfma" and a["peak"] == 2500.0 and a["unit"] == "TFLOP/s" and a["launches"] > 0, a
        assert abs(a["frac"] - a["achieved"] / a["peak"]) < 1e-9 and 0.02 < a["frac"] < 1.0, a
    assert "attn_fwd2_kernel" in ra["fwd"]["kernel"] and "attn_bwd_dq2_kernel" in ra["bwd"]["kernel"]
    m = d["merge"]
    assert m["roofline"]["bound"] == "hbm" and m["algorithmic_bytes"] == 1077239808 and 0.3 < m["roofline"]["frac"] < 1.0
    # parity gate beside the number: the buffer the merge leg has just timed, against the CPU oracle on layers 0 and 11
    assert d["parity"]["merge"] == "bit-exact", d["parity"]
    # the other BASELINE configs, measured in the same run (configs[2], [4], [3]'s task-vector / RegMean / Gram legs)
    sec = d["secondary"]
    for k in ("all_moe_b22", "irtr_ufo_b20", "task_vector", "regmean_base", "gram_capture_base"):
        assert k in sec and "error" not in sec[k], (k, sec.get(k))
    assert sec["all_moe_b22"]["samples_per_s"] > 50 and sec["irtr_ufo_b20"]["samples_per_s"] > 50
    assert 0.3 < sec["task_vector"]["frac_of_hbm_peak"] < 1.0
    assert 0.0 < sec["regmean_base"]["frac_of_fp64_mfma_peak"] < 1.0 and sec["regmean_base"]["seconds"] < 10
    assert all(0.0 < sec["gram_capture_base"][k]["frac_of_fp64_mfma_peak"] < 1.0 for k in ("D768", "D3072"))


def _run_bench(argv, env=None, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT, env=dict(os.environ, **(env or {})))
    return r


SMALL = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-calibrate", "--no-merge", "--batch", "2",
         "--image-size", "224"]


@pytest.mark.gpu
def test_bench_gpus2_launches_two_ranks_on_one_device():
    """`python bench.py --gpus 2` (no launcher, no WORLD_SIZE) must start two rank processes itself (reference:
    run.py:263-288, Lightning spawns one process per GPU).  The test box has one GPU: both ranks share it and talk gloo."""
    env = {"VLM_BENCH_ONE_DEVICE": "1", "VLM_DIST_BACKEND": "gloo"}
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + SMALL, capture_output=True,
                       text=True, timeout=900, cwd=ROOT, env=dict(env_clean, **env))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and abs(d["value"] - 4 / (d["ms_per_step"] / 1e3)) < 1e-6 * d["value"] + 1e-3
    assert "exposed_comm_ms_per_step" in d


@pytest.mark.gpu
def test_bench_irtr_task_at_two_ranks_with_secondary_legs_and_step_parity():
    """`--task irtr` is BASELINE configs[4]'s workload (run.py:263-288 + config.py:478-496): two ranks on the one device, the
    secondary data-parallel legs run by BOTH ranks (384^2 only, so this small run asks for none), and at N = 1 the configs[0]
    step-loss gate against the oracle."""
    env = {"VLM_BENCH_ONE_DEVICE": "1", "VLM_DIST_BACKEND": "gloo"}
    env_clean = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--task", "irtr"] + SMALL, capture_output=True,
                       text=True, timeout=900, cwd=ROOT, env=dict(env_clean, **env))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and "task_finetune_irtr_coco" in d["config"]["workload"] and "configs[4]" in d["config"]["workload"]
    assert d["value"] > 0 and d["config"]["global_batch"] == 4
    # the other two data-parallel workloads, run by both ranks after the headline (here at the small geometry)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--task", "irtr"] + [a for a in SMALL if a != "--no-secondary"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env_clean, VLM_BENCH_FORCE_SECONDARY="1", **env))
    assert r.returncode == 0, r.stderr[-3000:]
    sec = json.loads(r.stdout.strip().splitlines()[-1])["secondary"]
    for k in ("ufo_b22", "all_moe_b22"):
        assert k in sec and "error" not in sec[k] and sec[k]["n_gpus"] == 2 and sec[k]["samples_per_s"] > 0, (k, sec.get(k))
    assert "irtr_ufo_b20" not in sec  # the headline itself


@pytest.mark.gpu
def test_bench_step_loss_gate_against_the_oracle():
    """The JSON line carries |loss_gpu - loss_oracle| of one configs[0] step (the oracle step is the CPU baseline's own)."""
    r = _run_bench(["--steps", "2", "--warmup", "1", "--no-calibrate", "--no-merge", "--no-secondary", "--batch", "2",
                    "--image-size", "224"], timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    p = d["parity"]
    assert p["step_loss_ok"] is True and p["step_loss_err"] <= p["step_loss_tol"], p
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and str(cb["cores"]) in cb["thread_sweep_seconds"]
    assert not any(k.startswith("_") for k in cb)


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{"VLM_SHARDED": "0"}, {"VLM_SHARDED": "1"},
                                 {"VLM_GRAD_COMM": "bf16", "VLM_GRAD_COLLECTIVE": "rs_ag"}])
def test_bench_rccl_branch_at_world_one(env):
    """VLM_BENCH_FORCE_DIST=1: the nccl (= RCCL) process group, the reducer's collectives (all-reduce; with VLM_SHARDED=1
    reduce_scatter_tensor / all_gather_into_tensor; with VLM_GRAD_COMM=bf16 + VLM_GRAD_COLLECTIVE=rs_ag the bf16 wire
    buffer through reduce-scatter + all-gather) and the barrier-fenced timing run on the one device."""
    r = _run_bench(SMALL, dict(env, VLM_BENCH_FORCE_DIST="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["grad_comm"]["dtype"] == ("bf16" if env.get("VLM_GRAD_COMM") == "bf16" else "fp32")
    assert d["grad_comm"]["collective"] == env.get("VLM_GRAD_COLLECTIVE", "allreduce")
