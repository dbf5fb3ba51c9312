"""bench.py's own rank launcher (`--gpus N` without a launcher), dry run: rendezvous + one all-reduce over gloo on the CPU.
Reference: run.py:263-288 (`gpus=N, accelerator="ddp"`: one process per GPU)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    e.update(kw)
    return e


def test_gpus2_spawns_two_ranks_and_relays_rank0_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3"], capture_output=True,
                       text=True, timeout=300, cwd=ROOT, env=_env(VLM_BENCH_DRY_RUN="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, lines  # ONE JSON line on stdout, everything else on stderr
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry_run"] is True


def test_every_dp_config_of_baseline_has_a_command_line():
    """BASELINE.json configs[4] is `--gpus 8 --task irtr` (per-GPU batch 20 by default), configs[2] `--arch all_moe`: the
    launcher hands the task through to its ranks (dry run at two ranks)."""
    for argv, want in ((["--task", "irtr"], ("task_finetune_irtr_coco", "ufo", 20, 4)),
                       (["--arch", "all_moe"], ("task_mlm_itm_ifm", "all_moe", 22, 2)),
                       ([], ("task_mlm_itm_ifm", "ufo", 22, 1))):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + argv, capture_output=True,
                           text=True, timeout=300, cwd=ROOT, env=_env(VLM_BENCH_DRY_RUN="1"))
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads(r.stdout.strip().splitlines()[-1])
        assert (d["task"], d["arch"], d["per_gpu_batchsize"], d["baseline_config"]) == want and d["n_gpus"] == 2


def test_gpus_must_agree_with_world_size():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=_env(VLM_BENCH_DRY_RUN="1", WORLD_SIZE="1", RANK="0"))
    assert r.returncode != 0 and "must agree" in r.stderr


def test_failing_rank_fails_the_launcher():
    # rank 1 is told a world size that disagrees with --gpus: it exits non-zero, the parent must too
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=_env(VLM_BENCH_DRY_RUN="1", VLM_BENCH_TEST_FAIL_RANK="1"))
    assert r.returncode != 0


def test_hung_ranks_are_ended_by_the_wall_clock_limit_and_stderr_is_tagged():
    # all ranks alive, none progressing (a collective that never completes): the launcher's own limit ends the job
    # (20 s: a rank's `import torch` takes longer than 3 s on a loaded box, and the tags below need the ranks to have started)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=_env(VLM_BENCH_DRY_RUN="1", VLM_BENCH_TEST_HANG="1", VLM_BENCH_TIMEOUT_S="20"))
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
    assert "VLM_BENCH_TIMEOUT_S" in r.stderr
    assert "[rank 0] hanging on purpose" in r.stderr and "[rank 1] hanging on purpose" in r.stderr
