#!/bin/bash
# per-kernel times of the standalone harness: tools/scratch/trace_attn.sh <binary> <args...>
cd /tmp && export TMPDIR=/tmp
BIN=$1; shift
OUT=/tmp/trace_attn_$$
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- $GRAFT_REPO_ROOT/tools/scratch/$BIN "$@" > $OUT.log 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/run_kernel_stats.csv")):
    print("%-60s calls %4s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
