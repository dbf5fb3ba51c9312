#!/bin/bash
# launch table + A/B of the step with this tree's library (one box)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
timeout 600 python tools/count_launches.py > $O/launches.txt 2>&1; grep -B1 -A22 "launches per step" $O/launches.txt | tail -30
for i in 1 2; do python bench.py --steps 12 --warmup 3 --no-merge --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['ms_per_step_median'], d['value'])"; done
