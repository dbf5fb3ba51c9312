#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1700 python -m pytest tests/test_model_gpu.py -m gpu -q > $O/tests_a.log 2>&1; tail -12 $O/tests_a.log | cut -c1-400
