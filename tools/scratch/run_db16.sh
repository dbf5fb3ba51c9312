#!/bin/bash
cd tools/scratch
for B in 88 66 22; do
echo "B=$B no dbias:"; timeout 120 ./attn_bench_FULL $B 0 1 1 0 | tail -1
for g in 1 2 3 4 5 6 8; do echo -n "g=$g: "; VLM_ATT_DB_GROUPS=$g timeout 120 ./attn_bench_FULL $B 0 1 1 1 | grep "^bwd"; done
done
