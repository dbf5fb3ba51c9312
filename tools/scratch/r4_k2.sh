#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_k2; mkdir -p $O
python -m pytest tests/test_model_gpu.py tests/test_ckpt_flow_gpu.py tests/test_ddp_losses_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary > $O/bench_ufo.json 2>$O/bench_err.log
python -c "import json;d=json.loads(open('$O/bench_ufo.json').read().strip().splitlines()[-1]);print('ufo', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
