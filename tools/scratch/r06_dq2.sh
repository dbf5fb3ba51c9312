#!/bin/bash
# forward / dQ: hand-placed kernels against the round-3 kernels: harness per-kernel times, stamps, then the attention tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for args in "88 0 1 1 1" "22 0 1 1 1"; do
  for v in 0 1; do
    echo -n "== DQ2=$v $args: "; VLM_ATT_DQ2=$v timeout 120 bash tools/scratch/trace_attn.sh attn_bench $args 2>&1 | grep -E "attn_bwd_dq" | awk '{print $1, $(NF-1), $NF}'
  done
done 2>&1 | tee $O/dq2_harness.txt
for args in "88 0 1 0" "22 0 1 0"; do
  for v in 0 1; do
    echo -n "== FWD2=$v $args: "; VLM_ATT_FWD2=$v timeout 120 bash tools/scratch/trace_attn.sh attn_bench $args 2>&1 | grep -E "attn_fwd" | awk '{print $1, $(NF-1), $NF}'
  done
done 2>&1 | tee -a $O/dq2_harness.txt
timeout 60 tools/scratch/attn_bench_diag 88 0 1 1 1 2>&1 | grep -E "wave 0" | cut -c1-300 | tee -a $O/dq2_harness.txt
timeout 60 tools/scratch/attn_bench_diag 88 0 1 0 2>&1 | grep -E "wave 0" | cut -c1-300 | tee -a $O/dq2_harness.txt
timeout 900 python -m pytest tests/test_attention_gpu.py -m gpu -q > $O/dq2_tests.log 2>&1; tail -4 $O/dq2_tests.log | cut -c1-300
