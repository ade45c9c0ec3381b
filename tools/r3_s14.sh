#!/bin/bash
mkdir -p gpurun_out
MARK=compact_plan_kernel CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs --steps 100 --warmup 20 --n-items 2000000 --seq-len 200 --negatives 1000 --loss softmax --batch 128" TAILN=120 bash tools/timeline.sh > gpurun_out/s14_timeline_c3.txt 2>&1
sed -n '/main queue/,$p' gpurun_out/s14_timeline_c3.txt
