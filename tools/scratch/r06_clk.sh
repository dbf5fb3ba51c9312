#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 1 0; do echo "== DQ2=$v"; VLM_ATT_DQ2=$v VLM_ATT_BWD_FUSED=1 timeout 60 tools/scratch/attn_bench_diag 88 0 1 1 1 2>&1 | grep -E "wave 0|bwd B" | cut -c1-700; done
