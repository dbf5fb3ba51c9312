#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_attention_gpu.py -m gpu -q 2>&1 | tail -3
for rep in 1 2; do for b in attn_bench_old attn_bench; do
  echo -n "$b fwd joint: "; timeout 60 tools/scratch/$b 88 0 1 0 2>&1 | grep -v occupancy
  echo -n "$b fwd sep: "; timeout 60 tools/scratch/$b 88 1 1 0 2>&1 | grep -v occupancy
  echo -n "$b bwd joint fused: "; timeout 60 tools/scratch/$b 88 0 1 1 1 2>&1 | grep -v "occupancy\|checksum"
  echo -n "$b bwd joint no-dbias: "; timeout 60 tools/scratch/$b 88 0 1 1 0 2>&1 | grep -v "occupancy\|checksum"
done; done
