#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_regmean; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_regmean.py > $O/log.txt 2>&1
tail -2 $O/log.txt
python3 - <<PY
import csv, glob
f = glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:12]:
    print(r["Name"][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
