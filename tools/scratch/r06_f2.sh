#!/bin/bash
# forward attention: hand-placed kernel (VLM_ATT_FWD2=1) against the round-3 kernel: tests, then harness times
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for args in "88 0 1 0" "22 0 1 0" "88 1 1 0"; do
  for v in 0 1; do
    echo -n "== FWD2=$v $args: "; VLM_ATT_FWD2=$v timeout 120 bash tools/scratch/trace_attn.sh attn_bench $args 2>&1 | grep -E "attn_fwd" | awk '{print $1, $(NF-1), $NF}'
  done
done 2>&1 | tee $O/f2_harness.txt
VLM_ATT_FWD2=1 timeout 60 tools/scratch/attn_bench_diag 88 0 1 0 2>&1 | tail -4 | tee $O/f2_diag.txt
VLM_ATT_FWD2=1 timeout 900 python -m pytest tests/test_attention_gpu.py -m gpu -q > $O/f2_tests.log 2>&1; tail -15 $O/f2_tests.log | cut -c1-300
