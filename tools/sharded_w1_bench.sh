#!/bin/bash
# Overhead of the multi-GPU step's code path itself at world 1 (no collectives): ShardedSparseDenseAdam (the optimizer the Trainer builds
# under torch.distributed: fixed-capacity row exchange, owner-side plan / reduction, flags) against the plain optimizer, same harness,
# same batches, interleaved.   usage (GPU box): bash tools/sharded_w1_bench.sh [reps]  -> "plain|sharded_w1 ms/step loss"
reps=${1:-3}
for rep in $(seq $reps); do
  for mode in plain sharded_w1; do
    flag=""; [ $mode = sharded_w1 ] && flag="--sharded-w1"
    python bench.py --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 200 --warmup 30 $flag 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', j['ms_per_step'], j['final_loss'])"
  done
done
