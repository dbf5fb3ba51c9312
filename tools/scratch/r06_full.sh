#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x --deselect tests/test_bench_gpu.py::test_bench_json_contract > $O/tests_full.log 2>&1; tail -6 $O/tests_full.log | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline > $O/bench_mid.json 2> $O/bench_mid.err; python -c "
import json;d=json.loads(open('$O/bench_mid.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['frac'],d.get('roofline_attention'),d['roofline'].get('traffic_source'))"
