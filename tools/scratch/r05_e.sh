#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_attention_gpu.py tests/test_ddp_gpu.py tests/test_ddp_losses_gpu.py "tests/test_model_gpu.py::test_train_mode_step_and_optimizer" -m gpu -q > $O/e_tests.log 2>&1; tail -12 $O/e_tests.log
VLM_ATT_BWD_FUSED=1 timeout 120 tools/scratch/attn_bench 88 0 1 1 1 2>&1 | grep -v occupancy
bash tools/scratch/pmc_attn.sh attn_bench 88 0 1 1 1 > $O/e_pmc1.txt 2>&1; grep dkvb $O/e_pmc1.txt
bash tools/scratch/pmc_attn2.sh attn_bench 88 0 1 1 1 > $O/e_pmc2.txt 2>&1; grep dkvb $O/e_pmc2.txt
