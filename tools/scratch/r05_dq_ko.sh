#!/bin/bash
# dQ kernel knock-outs (timing only, results wrong): 1 no exp, 2 no dQ MFMAs, 3 no bias, 4 no staging / barrier
cd $GRAFT_REPO_ROOT
for bin in attn_bench_base attn_bench_ko1 attn_bench_ko2 attn_bench_ko3 attn_bench_ko4; do
  echo -n "== $bin: "; bash tools/scratch/trace_attn.sh $bin 88 0 1 1 1 2>&1 | grep -E "dq_kernel" | awk '{print $(NF-1), $NF}'
done
