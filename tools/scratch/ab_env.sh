#!/bin/bash
# A/B of an environment switch inside the training step on one box, alternating: tools/scratch/ab_env.sh VAR  (VAR=0 against VAR=1)
cd $GRAFT_REPO_ROOT
V=$1
for v in 0 1 0 1; do
  echo -n "$V=$v: "; env $V=$v timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['ms_per_step_median'], d['value'], d['config']['final_loss'])"
done
