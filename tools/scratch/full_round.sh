#!/bin/bash
# everything the round's numbers come from, one box: GPU tests, smoke, the bench line, the profile passes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r04; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_line.json 2> $O/bench_err.log; tail -c 1500 $O/bench_line.json
bash tools/profile_round.sh r04 > $O/profile.log 2>&1; tail -30 $O/profile.log
