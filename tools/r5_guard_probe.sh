#!/bin/bash
# round 5: id guard tests + the touched suites, the LDS-DMA probe, a quick A/B that the guard's extra kernel argument costs nothing
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_id_guard_gpu.py -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r5_guard_tests.txt
cat gpurun_out/r5_guard_tests.txt
timeout 1500 python -m pytest tests/test_sharded.py tests/test_gpu_parity.py tests/test_trainer_gpu.py tests/test_full_size_gpu.py tests/test_distributed_trainer.py -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r5_guard_suites.txt
cat gpurun_out/r5_guard_suites.txt
timeout 300 tools/probe/bin/ldsdma_probe > gpurun_out/r5_ldsdma_probe.txt 2>&1
cat gpurun_out/r5_ldsdma_probe.txt
for rep in 1 2; do
  timeout 600 python bench.py --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 200 --warmup 30 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tree', j['ms_per_step'], j['final_loss'])"
done > gpurun_out/r5_guard_bench.txt 2>&1
cat gpurun_out/r5_guard_bench.txt
