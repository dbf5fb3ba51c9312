#!/bin/bash
# kernel-trace stats of a short bench run: usage tools/prof_step.sh TAG [env assignments...]
TAG=$1; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o run -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-calibrate --no-merge $BENCH_ARGS > $OUT/trace.log 2>&1
cd $ROOT
python3 tools/prof_summary.py $OUT/trace/run_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv "bench.py --steps 4 --warmup 2 under rocprofv3 --kernel-trace --stats ($*)"
head -22 $OUT/${TAG}_kernel_stats.csv
