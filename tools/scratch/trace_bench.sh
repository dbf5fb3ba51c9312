#!/bin/bash
# kernel trace of a short bench run, top kernels by total time -> gpurun_out/trace_bench/
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/trace_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o run -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-calibrate > $OUT/log.txt 2>&1
cd $ROOT
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/t/**/run_kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("total kernel ms per step (6 steps): %.2f" % (tot / 6e6))
    for r in rows[:24]:
        print("%-90s calls %6s avg %9.1f us  %6.2f ms/step  %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 6e6, float(r["Percentage"])))
PY
tail -1 $OUT/log.txt | cut -c1-200
