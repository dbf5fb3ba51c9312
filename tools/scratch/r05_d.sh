#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
: > $O/attn_harness_d.txt
for B in 88 22; do for mode in 0 1; do for f in 0 1 1; do
  echo -n "B=$B mode=$mode fused=$f : " >> $O/attn_harness_d.txt
  VLM_ATT_BWD_FUSED=$f timeout 120 tools/scratch/attn_bench $B $mode 1 1 1 2>&1 | grep -v occupancy | tr '\n' ' ' >> $O/attn_harness_d.txt; echo >> $O/attn_harness_d.txt
done; done; done
cat $O/attn_harness_d.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > $O/gputest_d.log 2>&1; tail -40 $O/gputest_d.log
cp gpurun_out/parity_errors.json $O/parity_errors_d.json
for v in "0 0" "1 1" "0 0" "1 1"; do
  set -- $v
  VLM_FOLD_LAYERSCALE=$1 VLM_ATT_BWD_FUSED=$2 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-merge --no-calibrate --no-secondary > /tmp/b.json 2>/tmp/b.err
  python -c "import json,sys;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('fold=$1 attfused=$2', round(d['value'],2), round(d['ms_per_step'],3), round(d['roofline']['achieved'],1))" | tee -a $O/d_ab.txt
done
timeout 900 python tools/contention_sweep.py > $O/contention.log 2>&1; tail -30 $O/contention.log
