#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_rowchain_gpu.py tests/test_trainer_gpu.py tests/test_catchup_ahead_gpu.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/s5_tests.txt
cat gpurun_out/s5_tests.txt
{
timeout 900 bash tools/ab_env.sh UR_SASREC_CHAIN "57 59 63 61" 2
} > gpurun_out/s5_ab.txt 2>&1
awk '{print $1,$2,$3}' gpurun_out/s5_ab.txt
