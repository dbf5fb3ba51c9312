#!/bin/bash
# A/B of two library builds inside the training step on one box: tools/scratch/ab/libvlm_hip_old.so (a build of an earlier tree,
# see docs/experiments.md) against the tree's own library, alternating; then the attention kernels' times from a serial kernel trace.
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/tools/scratch/ab/libvlm_hip_old.so
for v in old new old new; do
  echo -n "$v: "; if [ $v = old ]; then export VLM_LIB_PATH=$OLD; else unset VLM_LIB_PATH; fi
  timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], d['value'])"
done
cd /tmp && export TMPDIR=/tmp
for v in old new; do
if [ $v = old ]; then export VLM_LIB_PATH=$OLD; else unset VLM_LIB_PATH; fi
rm -rf /tmp/tr$v; VLM_BENCH_SETUP_STEPS=0 VLM_WGRAD_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$v -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer > /tmp/tr$v.log 2>&1
python3 - <<PY
import csv
print("$v (6 steps, every launch alone on the chip)")
for r in csv.DictReader(open("/tmp/tr$v/run_kernel_stats.csv")):
    if "attn" in r["Name"]: print("  %-50s calls %4s  %8.3f ms/step  avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["TotalDurationNs"])/6e6, float(r["AverageNs"])/1e3))
PY
done
