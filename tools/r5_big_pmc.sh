#!/bin/bash
# PMC evidence for the 128 x 128 LDS-DMA weight-gradient kernel (UR_TN_BIG=1) against the 64 x 64 one: HBM bytes, MFMA busy, L2 requests
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for big in 1 0; do
  export UR_TN_BIG=$big
  bash tools/pmc_hbm.sh r5_big${big} 2>&1 | grep -E "gemm_tn|reduce_batch"
  PMC_FILTER="gemm_tn" bash tools/pmc_groups.sh r5_big${big} "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
done > gpurun_out/r5_big_pmc.txt 2>&1
cat gpurun_out/r5_big_pmc.txt
