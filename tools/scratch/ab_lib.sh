#!/bin/bash
# A/B of two builds of the library on one box: tools/scratch/ab/libvlm_old.so against the in-tree one (bench.py ufo, alternating)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_ab; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_attention_gpu.py tests/test_model_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/scratch/persist_check.py 2>&1 | tail -2
: > $O/gemm_ab.txt
for a in "54296 3072 768 5" "13574 3072 768 5" "54296 2304 768 0" "54296 3072 768 1" "54296 768 3072 3" "54296 768 768 3" "54296 768 3072 0" "54296 768 768 0" "54296 768 2304 0"; do
  for b in head new head new; do
    echo -n "$b $a : " >> $O/gemm_ab.txt
    timeout 120 tools/scratch/gemm_bench_$b $a 2>&1 | cut -c1-60 >> $O/gemm_ab.txt
  done
done
cat $O/gemm_ab.txt
for v in old new old new old new; do
  if [ $v = old ]; then export VLM_LIB_PATH=$GRAFT_REPO_ROOT/tools/scratch/ab/libvlm_old.so; else unset VLM_LIB_PATH; fi
  python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary > /tmp/b.json 2>/tmp/b.err
  python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('ufo $v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])" | tee -a $O/bench_ab.txt
done
