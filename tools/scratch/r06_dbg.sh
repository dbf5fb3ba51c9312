#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "2 0"; do echo "=== case $a"; VLM_ATT_FWD2=1 timeout 120 python tools/scratch/dbg_fwd2.py $a 2>&1 | grep -v amdgpu.ids | tail -6; done
