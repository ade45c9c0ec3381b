#!/bin/bash
# A/B of several environment COMBINATIONS on one box, interleaved: tools/ab_env2.sh <reps> "A=1,B=0" "A=1,B=1" ...  [-- extra bench args]
reps=$1; shift
specs=(); while [ -n "$1" ] && [ "$1" != "--" ]; do specs+=("$1"); shift; done; shift
for rep in $(seq $reps); do
  for sp in "${specs[@]}"; do
    env $(echo $sp | tr ',' ' ') python bench.py --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 200 --warmup 30 "$@" 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$sp', j['ms_per_step'], j['final_loss'])"
  done
done
