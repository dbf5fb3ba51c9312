#!/bin/bash
# dQ kernel: bias rows requested per block (early) against the end-of-trip request; per-kernel times of the harness
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05q; mkdir -p $O
timeout 600 python -m pytest tests/test_attention_gpu.py -m gpu -x -q > $O/dq_tests.log 2>&1; tail -2 $O/dq_tests.log
for bin in attn_bench_base attn_bench attn_bench_base attn_bench; do
  for args in "88 0 1 1 1" "22 1 1 1 1"; do
    echo "== $bin $args"; bash tools/scratch/trace_attn.sh $bin $args 2>&1 | grep -E "dq_kernel|dkvb"
  done
done
