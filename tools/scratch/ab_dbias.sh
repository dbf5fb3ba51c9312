#!/bin/bash
# A/B of the bias-table-gradient kernel: tools/scratch/ab/libvlm_old.so (previous build) against the in-tree library, same box
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4_dbias2
mkdir -p $OUT
python -m pytest tests/test_attention_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
for v in old new old new; do
  if [ $v = old ]; then export VLM_LIB_PATH=$GRAFT_REPO_ROOT/tools/scratch/ab/libvlm_old.so; else unset VLM_LIB_PATH; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$v -o run -- python3 $GRAFT_REPO_ROOT/tools/bench_attn.py 88 > $OUT/bench_$v.log 2>&1
  echo "== $v"; grep -E "attn_bwd_dbias16|attn_bwd_dq|attn_bwd_dkv" $OUT/t_$v/run_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
done
unset VLM_LIB_PATH
cd $GRAFT_REPO_ROOT
for v in old new old new; do
  if [ $v = old ]; then export VLM_LIB_PATH=$GRAFT_REPO_ROOT/tools/scratch/ab/libvlm_old.so; else unset VLM_LIB_PATH; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary --no-gemm-timer > /tmp/b.json 2>/tmp/b.err
  python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('ufo $v', d['value'], d['ms_per_step'])"
done
