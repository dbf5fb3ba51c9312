#!/bin/bash
# A/B of an environment switch on one box: tools/scratch/ab_env.sh VAR  (bench.py ufo with VAR=0 / VAR=1 alternating)
cd $GRAFT_REPO_ROOT
V=$1
O=$GRAFT_REPO_ROOT/gpurun_out/r4_ab; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q > $O/pytest_env.log 2>&1; tail -3 $O/pytest_env.log
for v in 0 1 0 1 0 1; do
  env $V=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer > /tmp/b.json 2>/tmp/b.err
  python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('ufo $V=$v', d['value'], d['ms_per_step'])" | tee -a $O/env_ab.txt
done
