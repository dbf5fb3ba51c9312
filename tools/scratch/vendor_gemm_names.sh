#!/bin/bash
# Which hipBLASLt solutions does torch pick at the training shapes?  (kernel names encode macro tile / staging choices)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/vendor_names
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o run -- python3 $ROOT/tools/bench_gemm.py > $OUT/log.txt 2>&1
cd $ROOT
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/t/**/run_kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:400], r["Calls"], r["AverageNs"])
PY
