#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_ops; mkdir -p $O
python tools/torch_ops_profile.py ufo > $O/torch_ops_ufo.txt 2>&1
cat $O/torch_ops_ufo.txt | tail -95
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "dense_bias or full_size" > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
