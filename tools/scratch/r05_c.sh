#!/bin/bash
# round 5: same-box baseline (the round-4 tree), then A/Bs of the two new switches, then the tests of the folded LayerScale
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
: > $O/c_ab.txt
line() { python -c "import json,sys;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print(sys.argv[1], round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), d['config']['final_loss'])" "$1" | tee -a $O/c_ab.txt; }
ARGS="--steps 12 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary"
for rep in 1 2; do
  (cd tools/scratch/r04_tree && timeout 300 python bench.py $ARGS > /tmp/b.json 2>/tmp/b.err); line "r04_tree"
  VLM_FOLD_LAYERSCALE=0 VLM_ATT_BWD_FUSED=0 timeout 300 python bench.py $ARGS > /tmp/b.json 2>/tmp/b.err; line "r05 fold=0 attfused=0"
  VLM_FOLD_LAYERSCALE=0 VLM_ATT_BWD_FUSED=1 timeout 300 python bench.py $ARGS > /tmp/b.json 2>/tmp/b.err; line "r05 fold=0 attfused=1"
  VLM_FOLD_LAYERSCALE=1 VLM_ATT_BWD_FUSED=1 timeout 300 python bench.py $ARGS > /tmp/b.json 2>/tmp/b.err; line "r05 fold=1 attfused=1"
done
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_f64_gpu.py tests/test_model_gpu.py tests/test_ddp_gpu.py -m gpu -x -q --durations=8 > $O/c_tests.log 2>&1; tail -25 $O/c_tests.log
timeout 600 python bench.py --steps 8 --warmup 2 > $O/bench_c.json 2> $O/bench_c.err; echo "bench rc=$?"; grep "^\[bench" $O/bench_c.err | tail -30
