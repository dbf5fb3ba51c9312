#!/bin/bash
cd $GRAFT_REPO_ROOT
for f in 1 0 1 0; do
VLM_FUSED_LOSS=$f python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer > /tmp/b.json 2>/tmp/b.err
python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('fused_loss=$f', d['value'], d['ms_per_step'])"
done
