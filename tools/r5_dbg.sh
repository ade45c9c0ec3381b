#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_id_guard_gpu.py -q -m gpu -x 2>&1 | tail -60 > gpurun_out/r5_guard_tests.txt
cat gpurun_out/r5_guard_tests.txt
timeout 600 python -m pytest tests/test_distributed_trainer.py -q -m gpu -x -k "second_table" 2>&1 | tail -8 > gpurun_out/r5_second_table.txt
cat gpurun_out/r5_second_table.txt
bash tools/r5_tn_ab.sh
