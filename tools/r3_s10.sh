#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sharded.py tests/test_distributed_trainer.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/s10_tests.txt
cat gpurun_out/s10_tests.txt
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs --steps 200 --warmup 20 --sharded-w1" TAILN=100 bash tools/timeline.sh > gpurun_out/s10_timeline_w1.txt 2>&1
sed -n '/main queue/,$p' gpurun_out/s10_timeline_w1.txt
