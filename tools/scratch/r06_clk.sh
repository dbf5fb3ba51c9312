#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 1 0 1; do echo "== FWD2=$v"; VLM_ATT_FWD2=$v timeout 60 tools/scratch/attn_bench_diag 88 0 1 0 2>&1 | tail -4; done
