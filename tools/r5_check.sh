#!/bin/bash
# round 5: the new multi-GPU robustness pieces on the GPU box -- loopback transport tests, id guard, the ladder with real workers
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_loopback_gpu.py -q -m gpu -x 2>&1 | tail -40 > gpurun_out/r5_loopback_tests.txt
cat gpurun_out/r5_loopback_tests.txt
timeout 900 python -m pytest tests/test_id_guard_gpu.py tests/test_catchup_ahead_gpu.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r5_guard_tests.txt
cat gpurun_out/r5_guard_tests.txt
timeout 1500 python -m pytest tests/test_distributed_trainer.py tests/test_sharded.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r5_dist_tests.txt
cat gpurun_out/r5_dist_tests.txt
timeout 1200 python -m pytest tests/test_bench_ladder.py -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r5_ladder_gpu.txt
cat gpurun_out/r5_ladder_gpu.txt
