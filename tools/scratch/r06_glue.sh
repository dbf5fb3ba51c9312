#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python tools/count_launches.py > $O/launches.txt 2>&1; grep -A22 "launches per step" $O/launches.txt | head -28
timeout 600 python tools/glue_by_line.py > $O/glue_by_line.txt 2>&1; grep -v "record_stream\|new_empty" $O/glue_by_line.txt | tail -60
