#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ce; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "cross_entropy or l2_normalize" > $O/pytest_k.log 2>&1; echo "rc=$?" >> $O/pytest_k.log; tail -12 $O/pytest_k.log
python -m pytest tests/test_model_gpu.py tests/test_ddp_losses_gpu.py -x -q -m gpu > $O/pytest_m.log 2>&1; echo "rc=$?" >> $O/pytest_m.log; tail -6 $O/pytest_m.log
python tools/count_launches.py ufo 2>&1 | grep "launches per step"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary > $O/bench_ufo.json 2>$O/bench_err.log
python -c "import json;d=json.loads(open('$O/bench_ufo.json').read().strip().splitlines()[-1]);print('ufo', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
