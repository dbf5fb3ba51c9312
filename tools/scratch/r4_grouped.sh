#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_grouped; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "grouped" > $O/pytest_grouped.log 2>&1; echo "rc=$?" >> $O/pytest_grouped.log
tail -15 $O/pytest_grouped.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "all_moe" > $O/pytest_model.log 2>&1; echo "rc=$?" >> $O/pytest_model.log
tail -8 $O/pytest_model.log
for g in 1 0 1 0; do
  VLM_GROUPED_GEMM=$g python bench.py --arch all_moe --steps 8 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate > $O/bench_all_moe_g$g.json 2>$O/bench_err.log
  python -c "import json;d=json.loads(open('$O/bench_all_moe_g$g.json').read().strip().splitlines()[-1]);print('grouped=$g', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate > $O/bench_ufo.json 2>>$O/bench_err.log
python -c "import json;d=json.loads(open('$O/bench_ufo.json').read().strip().splitlines()[-1]);print('ufo', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
