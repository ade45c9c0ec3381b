#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s8_tests.txt
cat gpurun_out/s8_tests.txt
python bench.py --no-cpu-baseline --no-gather-bench --no-extra-legs --steps 200 --warmup 30 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['mfma_classes_warmup'])"
