#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_full; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
tail -25 $O/pytest_gpu.log
