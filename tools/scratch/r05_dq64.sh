#!/bin/bash
# dQ kernel: 64 rows per wave (VLM_ATT_DQ64=1) against the 32-row kernel: attention tests, then per-kernel times of the harness
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
VLM_ATT_DQ64=1 timeout 600 python -m pytest tests/test_attention_gpu.py -m gpu -x -q > $O/dq64_tests.log 2>&1; tail -3 $O/dq64_tests.log
for v in 0 1 0 1; do
  for args in "88 0 1 1 1" "22 1 1 1 1" "88 1 1 1 1"; do
    echo -n "== DQ64=$v $args: "; VLM_ATT_DQ64=$v bash tools/scratch/trace_attn.sh attn_bench $args 2>&1 | grep -E "dq" | awk '{print $(NF-1), $NF}'
  done
done
