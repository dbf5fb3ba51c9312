#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
for v in 1 0; do
  VLM_FOLD_LAYERSCALE=$v timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -k "golden or injected" > $O/i_tests_$v.log 2>&1; tail -3 $O/i_tests_$v.log
  cp gpurun_out/parity_errors.json $O/parity_fold$v.json
done
python - <<PY
import json
a=json.load(open("$O/parity_fold1.json")); b=json.load(open("$O/parity_fold0.json"))
for t in sorted(a):
    for k in sorted(a[t]):
        if "grad" in k or "logits" in k or k in ("cls_feats",):
            print("%-70s %-40s fold1 %s fold0 %s" % (t[:70], k[:40], a[t][k] if not isinstance(a[t][k], list) else "...", b.get(t, {}).get(k) if not isinstance(b.get(t, {}).get(k), list) else "..."))
PY
