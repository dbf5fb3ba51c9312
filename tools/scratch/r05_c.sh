#!/bin/bash
# round 5: folded LayerScale -- kernel tests, model goldens, the two-rank reducer tests, then an A/B of the switch on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_ddp_gpu.py tests/test_attention_gpu.py -m gpu -x -q --durations=8 > $O/c_tests.log 2>&1; tail -22 $O/c_tests.log
for v in 0 1 0 1; do
  VLM_FOLD_LAYERSCALE=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary > /tmp/b.json 2>/tmp/b.err
  python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('fold=$v', round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), d['config']['final_loss'])" | tee -a $O/c_ab.txt
done
for v in 0 1; do
  VLM_ATT_BWD_FUSED=$v python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary > /tmp/b.json 2>/tmp/b.err
  python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('attfused=$v', round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1), d['config']['final_loss'])" | tee -a $O/c_ab.txt
done
