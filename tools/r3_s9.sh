#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sharded.py tests/test_distributed_trainer.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/s9_tests.txt
cat gpurun_out/s9_tests.txt
bash tools/sharded_w1_bench.sh 2 > gpurun_out/s9_w1.txt 2>&1
cat gpurun_out/s9_w1.txt
