#!/bin/bash
# round 5: the fused dK/dV + bias-gradient kernel against the dkv + dbias16 pair: correctness first, then the harness timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_attention_gpu.py -m gpu -x -q > $O/attn_tests.log 2>&1; tail -5 $O/attn_tests.log
: > $O/attn_harness.txt
for B in 88 22; do for mode in 0 1; do for f in 0 1 0 1; do
  echo -n "B=$B mode=$mode fused=$f : " >> $O/attn_harness.txt
  VLM_ATT_BWD_FUSED=$f timeout 120 tools/scratch/attn_bench $B $mode 1 1 1 2>&1 | grep -v occupancy | tr '\n' ' ' >> $O/attn_harness.txt; echo >> $O/attn_harness.txt
done; done; done
for B in 88; do for mode in 0; do
  echo -n "B=$B mode=$mode no dbias : " >> $O/attn_harness.txt
  timeout 120 tools/scratch/attn_bench $B $mode 1 1 0 2>&1 | grep -v occupancy | tr '\n' ' ' >> $O/attn_harness.txt; echo >> $O/attn_harness.txt
done; done
cat $O/attn_harness.txt
