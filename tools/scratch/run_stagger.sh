#!/bin/bash
# stagger sweep over the training shapes (GPU box)
B=$GRAFT_REPO_ROOT/tools/scratch/gemm_stagger
O=$GRAFT_REPO_ROOT/gpurun_out/stagger.txt
: > $O
for a in "54296 3072 768 0" "54296 3072 768 1" "54296 3072 768 2" "54296 2304 768 0" "54296 768 768 3" "54296 768 2304 0" "54296 768 3072 3" "54296 768 3072 0" "13574 3072 768 1" "13574 768 3072 3"; do
  timeout 120 $B $a >> $O 2>&1
done
cat $O
