#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_misc; mkdir -p $O
python -m pytest tests/test_gram_regmean_gpu.py tests/test_f64_gpu.py tests/test_merge_gpu.py -x -q -m gpu > $O/pytest_f64.log 2>&1; echo "rc=$?" >> $O/pytest_f64.log; tail -5 $O/pytest_f64.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "dense_bias" > $O/pytest_dense.log 2>&1; echo "rc=$?" >> $O/pytest_dense.log; tail -30 $O/pytest_dense.log
python tools/bench_f64_leg.py 2>&1 | tail -3
