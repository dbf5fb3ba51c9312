#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
bash tools/scratch/r05_attn.sh
timeout 900 python bench.py --steps 8 --warmup 2 > $O/bench_b.json 2> $O/bench_b.err; echo "bench rc=$?"; grep "^\[bench" $O/bench_b.err | tail -40
timeout 900 python -m pytest tests/test_bench_gpu.py -m gpu -q --durations=10 > $O/bench_tests.log 2>&1; tail -25 $O/bench_tests.log
