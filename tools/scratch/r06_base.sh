#!/bin/bash
# round-6 baseline on one box: harness per-kernel times, then the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for args in "88 0 1 0" "88 1 1 0" "22 0 1 0" "88 0 1 1 1" "22 0 1 1 1"; do
  echo "== $args"; bash tools/scratch/trace_attn.sh attn_bench $args 2>&1 | grep -E "attn"
done > $O/base_harness.txt 2>&1
cat $O/base_harness.txt
timeout 900 python bench.py --no-cpu-baseline > $O/base_bench.json 2> $O/base_bench.err; tail -c 1500 $O/base_bench.json
