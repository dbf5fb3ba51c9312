#!/bin/bash
# the attention-harness block of tools/profile_round.sh alone (after rebuilding tools/scratch/attn_bench[_diag])
cd $GRAFT_REPO_ROOT
TAG=${1:-r06}; OUT=gpurun_out/$TAG; mkdir -p $OUT
{
  for args in "88 0 1 0" "22 0 1 0"; do for v in 0 1; do
    echo -n "forward  hand-placed=$v  B mode bias = $args: "; VLM_ATT_FWD2=$v bash tools/scratch/trace_attn.sh attn_bench $args 2>&1 | grep -E "attn_fwd" | awk '{print $1, $(NF-1), $NF}'
  done; done
  for args in "88 0 1 1 1" "22 0 1 1 1"; do for v in 0 1; do
    echo -n "backward hand-placed dQ=$v  $args: "; VLM_ATT_DQ2=$v bash tools/scratch/trace_attn.sh attn_bench $args 2>&1 | grep -E "attn_bwd" | awk '{printf "%s %s us; ", $1, $(NF-1)} END {print ""}'
  done; done
  for v in 0 1; do echo "stamps forward hand-placed=$v:"; VLM_ATT_FWD2=$v tools/scratch/attn_bench_diag 88 0 1 0 2>&1 | grep -E "wave 0 clock"; done
  for v in 0 1; do echo "stamps backward (dQ kernel) hand-placed=$v:"; VLM_ATT_DQ2=$v tools/scratch/attn_bench_diag 88 0 1 1 1 2>&1 | grep -E "wave 0 clock"; done
} > $OUT/${TAG}_attention_harness.txt 2>&1
cat $OUT/${TAG}_attention_harness.txt
