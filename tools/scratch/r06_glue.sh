#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python tools/count_launches.py > $O/launches.txt 2>&1; tail -45 $O/launches.txt
timeout 600 python tools/scratch/glue_by_line.py > $O/glue_by_line.txt 2>&1; tail -70 $O/glue_by_line.txt
