#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_run2; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "grouped" > $O/pytest_grouped.log 2>&1; echo "rc=$?" >> $O/pytest_grouped.log
tail -5 $O/pytest_grouped.log
python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_ufo.json 2>$O/bench_err.log; echo "bench rc=$?"
tail -c 3000 $O/bench_err.log
python - <<PY
import json
d=json.loads(open('$O/bench_ufo.json').read().strip().splitlines()[-1])
print('ufo', d['value'], d['ms_per_step'], d['roofline']['achieved'])
print(json.dumps(d.get('secondary'), indent=1))
PY
