#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
timeout 600 python -m pytest tests/test_attention_gpu.py -m gpu -q > $O/f_tests.log 2>&1; tail -4 $O/f_tests.log
for B in 88 22; do for mode in 0 1; do
  echo -n "B=$B mode=$mode fused=1 : "; VLM_ATT_BWD_FUSED=1 timeout 120 tools/scratch/attn_bench $B $mode 1 1 1 2>&1 | grep -v "occupancy\|checksum"
done; done
for env in "A=1" "VLM_FOLD_LAYERSCALE=0" "VLM_ATT_BWD_FUSED=0" "VLM_FOLD_LAYERSCALE=0 VLM_ATT_BWD_FUSED=0"; do
  echo "== $env"; env $env timeout 300 python -m pytest "tests/test_ddp_losses_gpu.py::test_gathering_losses_two_ranks[irtr]" -m gpu -q 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-300
done
