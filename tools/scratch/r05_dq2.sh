#!/bin/bash
cd $GRAFT_REPO_ROOT
for pad in 0 40960; do
  echo "== pad $pad"; VLM_DIAG_DQ_PAD_LDS=$pad bash tools/scratch/trace_attn.sh attn_bench_diag 88 0 1 1 1 2>&1 | grep -E "dq_kernel"
done
VLM_DIAG_DQ_PAD_LDS=40960 tools/scratch/attn_bench_diag 88 0 1 1 1 2>&1 | grep "wave 0"
