"""Merge kernel: grid size / cache-policy variants on this box (VLM_MERGE_VARIANT = blocks per CU, nt loads, nt stores).
One child process per variant (the switch is read once per process).  python tools/bench_merge_variants.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = "import sys; sys.path.insert(0, %r); import __graft_entry__ as ge; ge.import_package(); import importlib, json; " \
        "bm = importlib.import_module('vl_merging_amd.bench_merge'); print(json.dumps(bm.run(reps=30)))" % ROOT

if __name__ == "__main__":
    rows = []
    for bpc in [int(x) for x in (sys.argv[1].split(',') if len(sys.argv) > 1 else '4,8,16,32'.split(','))]:
        for ntl in ((1,) if len(sys.argv) > 1 else (1, 0)):
            for nts in ((1,) if len(sys.argv) > 1 else (1, 0)):
                env = dict(os.environ, VLM_MERGE_VARIANT="%d,%d,%d" % (bpc, ntl, nts))
                out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                if not line:
                    print("variant", env["VLM_MERGE_VARIANT"], "failed:", out.stderr[-300:])
                    continue
                r = json.loads(line[-1])
                rows.append((r["GBps"], bpc, ntl, nts, r["seconds_median"] * 1e6))
                print("blocks/CU %2d  nt loads %d  nt stores %d : %.1f GB/s (%.1f us)" % (bpc, ntl, nts, r["GBps"], r["seconds_median"] * 1e6), flush=True)
    best = max(rows)
    print("best: blocks/CU %d, nt loads %d, nt stores %d -> %.1f GB/s = %.3f of 8 TB/s" % (best[1], best[2], best[3], best[0], best[0] / 8000))
