#!/bin/bash
# SQ counters of the attention kernels in the harness: where the wave cycles go (MI355X_MICROARCH.md, rocprofv3 PMC slots)
cd /tmp && export TMPDIR=/tmp
ARGS=${1:-"88 0 1 1 1"}
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM"; do
  rm -rf /tmp/pa; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pa -o run -- $GRAFT_REPO_ROOT/tools/scratch/attn_bench $ARGS > /tmp/pa.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/pmc_report.py /tmp/pa attn_
done
