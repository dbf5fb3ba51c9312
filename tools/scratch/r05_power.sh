#!/bin/bash
# clocks / power while the training leg runs (is the step power-capped?)
O=gpurun_out/r05q; mkdir -p $O
rocm-smi --showpower --showclocks --showmaxpower > $O/smi_idle.txt 2>&1
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; sleep 0.5; done ) > $O/smi_load.txt 2>&1 &
SM=$!
timeout 300 python bench.py --steps 60 --warmup 4 --no-cpu-baseline --no-secondary > $O/power_bench.json 2> $O/power_bench.err
wait $SM
tail -c 600 $O/power_bench.json; echo; head -30 $O/smi_load.txt
