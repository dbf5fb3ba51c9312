#!/bin/bash
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r4_irtr; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o run -- python3 $GRAFT_REPO_ROOT/tools/irtr_step.py 6 > $O/log.txt 2>&1
grep "irtr ufo" $O/log.txt
cd $GRAFT_REPO_ROOT
python3 tools/prof_summary.py $O/trace/run_kernel_stats.csv $O/irtr_kernel_stats.csv "tools/irtr_step.py 6 (8 steps) under rocprofv3 --kernel-trace --stats"
head -30 $O/irtr_kernel_stats.csv
