#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 2 2 6; do
  python bench.py --steps 8 --warmup $w --no-cpu-baseline --no-secondary --no-merge --no-calibrate > /tmp/b.json 2>/dev/null
  python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('warmup $w', round(d['value'],1), round(d['ms_per_step'],2), d['step_ms_rank0'])"
done
python bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-secondary --no-merge --no-calibrate > /tmp/b.json 2>/dev/null
python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('steps 20', round(d['value'],1), round(d['ms_per_step'],2), d['step_ms_rank0'])"
