#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1700 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "l2norm or contrastive or small_cross or cross_entropy" > $O/tests_a.log 2>&1; tail -6 $O/tests_a.log | cut -c1-400
timeout 1700 python -m pytest tests/test_model_gpu.py tests/test_ddp_losses_gpu.py tests/test_ddp_gpu.py tests/test_run_gpu.py -m gpu -q -x > $O/tests_b.log 2>&1; tail -8 $O/tests_b.log | cut -c1-400
timeout 600 python tools/count_launches.py > $O/launches.txt 2>&1; grep -A18 "launches per step" $O/launches.txt | head -24
