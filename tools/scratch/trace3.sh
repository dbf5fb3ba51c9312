#!/bin/bash
# serial kernel traces of the step with two library builds (tools/scratch/ab/libvlm_hip_old.so = a build of an earlier tree / in-tree): per-kernel ms per step
cd /tmp && export TMPDIR=/tmp
for v in old intree; do
  case $v in
    old) export VLM_LIB_PATH=$GRAFT_REPO_ROOT/tools/scratch/ab/libvlm_hip_old.so;;
    intree) unset VLM_LIB_PATH;;
  esac
  rm -rf /tmp/tr$v; VLM_BENCH_SETUP_STEPS=0 VLM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$v -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer > /tmp/tr$v.log 2>&1
done
python3 - <<'PY'
import csv, re, collections
tabs = {}
for v in ("old", "intree"):
    d = collections.defaultdict(float)
    for r in csv.DictReader(open("/tmp/tr%s/run_kernel_stats.csv" % v)):
        k = re.sub(r"[<(].*", "", r["Name"]).replace("void ", "")
        d[k] += float(r["TotalDurationNs"]) / 6e6
    tabs[v] = d
keys = sorted(tabs["old"], key=lambda k: -tabs["old"][k])[:14]
print("%-28s %8s %8s" % ("ms per step", "old", "intree"))
for k in keys:
    print("%-28s %8.3f %8.3f" % (k[:28], tabs["old"][k], tabs["intree"].get(k, 0)))
print("%-28s %8.3f %8.3f" % ("total", *(sum(tabs[v].values()) for v in ("old", "intree"))))
PY
