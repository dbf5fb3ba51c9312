#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  echo -n "VLM_WGRAD_STREAM=$v : "; VLM_WGRAD_STREAM=$v python tools/irtr_step.py 12 2>&1 | grep "irtr ufo"
done
for v in 0 1 0 1; do
  VLM_WGRAD_STREAM=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer > /tmp/b.json 2>/tmp/b.err
  python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('ufo VLM_WGRAD_STREAM=$v', d['value'], d['ms_per_step'])"
done
