#!/bin/bash
# everything the round's numbers come from, one box: GPU tests, smoke, the bench line, the profile passes
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -q --durations=10 > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/${TAG}_parity_errors.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
python bench.py > $O/bench_line.json 2> $O/bench_err.log; tail -c 1500 $O/bench_line.json
bash tools/profile_round.sh $TAG > $O/profile.log 2>&1; tail -30 $O/profile.log
python tools/bench_gemm.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_gemm_vs_hipblaslt_m13574.txt
python tools/bench_gemm.py 54296 2>/dev/null | grep -v amdgpu > $O/${TAG}_gemm_vs_hipblaslt_m54296.txt
VLM_GEMM_TAIL_SPLIT=1 python tools/bench_gemm.py 54296 2>/dev/null | grep -v amdgpu > $O/${TAG}_gemm_vs_hipblaslt_m54296_tail_split.txt
tail -q -n 1 $O/${TAG}_gemm_vs_hipblaslt_m13574.txt $O/${TAG}_gemm_vs_hipblaslt_m54296.txt $O/${TAG}_gemm_vs_hipblaslt_m54296_tail_split.txt
