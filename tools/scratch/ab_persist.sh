#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_persist; mkdir -p $O
python tools/scratch/persist_check.py > $O/check.log 2>&1; tail -8 $O/check.log
: > $O/ab.txt
for a in "54296 2304 768 0" "54296 768 768 3" "54296 3072 768 1" "54296 768 3072 3" "54296 3072 768 2" "54296 768 3072 0" "54296 768 768 0" "13574 2304 768 0" "13574 768 768 3" "13574 3072 768 1" "13574 768 3072 3" "13574 3072 768 2" "4096 4096 4096 0"; do
  for b in base persist base persist; do
    echo -n "$b $a : " >> $O/ab.txt
    timeout 120 tools/scratch/gemm_bench_$b $a 2>&1 | cut -c1-60 >> $O/ab.txt
  done
done
cat $O/ab.txt
