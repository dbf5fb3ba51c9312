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
    m = d["merge"]
    assert m["roofline"]["bound"] == "hbm" and m["algorithmic_bytes"] == 1077239808 and 0.3 < m["roofline"]["frac"] < 1.0
