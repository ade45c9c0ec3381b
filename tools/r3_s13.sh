#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q -k "gru or GRU or g7 or c4 or C4" 2>&1 | tail -8 > gpurun_out/s13_tests.txt
cat gpurun_out/s13_tests.txt
timeout 1200 python bench.py --all-configs --steps 20 --warmup 5 --no-cpu-baseline --no-gather-bench --no-extra-legs 2>/dev/null | tail -1 > gpurun_out/s13_bench_all.json
python - <<'P'
import json
j=json.loads(open('gpurun_out/s13_bench_all.json').read())
for k,v in j['other_configs'].items():
    print(k, v['ms_per_step'], v['roofline']['kernel'], v['roofline']['frac'], v['kernel_time_ms_per_step'])
P
