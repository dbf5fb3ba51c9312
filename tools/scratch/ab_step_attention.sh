#!/bin/bash
# A/B of the hand-placed attention kernels (forward + dQ) inside the training step (same box, alternating): ms per step,
# then the attention kernels' per-launch times from a kernel trace of each arm
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for v in 0 1 0 1; do
  echo -n "hand-placed=$v: "; VLM_ATT_FWD2=$v VLM_ATT_DQ2=$v timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done 2>&1 | tee $O/ab_step_attn.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
VLM_BENCH_SETUP_STEPS=0 VLM_WGRAD_STREAM=0 VLM_ATT_FWD2=$v VLM_ATT_DQ2=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr$v -o run -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer > /tmp/tr$v.log 2>&1
python3 - <<PY
import csv
print("hand-placed=$v (6 steps, every launch alone on the chip)")
for r in csv.DictReader(open("/tmp/tr$v/run_kernel_stats.csv")):
    if "attn" in r["Name"]: print("  %-50s calls %4s  %8.3f ms/step  avg %8.1f us" % (r["Name"][:50], r["Calls"], float(r["TotalDurationNs"])/6e6, float(r["AverageNs"])/1e3))
PY
done 2>&1 | tee -a $GRAFT_REPO_ROOT/$O/ab_step_attn.txt
