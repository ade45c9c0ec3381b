#!/bin/bash
# A/B of one environment switch on the same box: tools/ab_env.sh VAR "v1 v2 ..." [reps] [extra bench args]
# prints VAR=value, ms/step, final loss for each run of the default bench (200 steps), interleaved so box drift hits both sides alike
# (build first and CHECK that it succeeded -- `python -c "import __graft_entry__ as g; g.build(); print('BUILD_OK')"` -- a failed build
# raises, but `... | tail -1 && gpurun ...` hides the exit code and the GPU box then runs the previous .so)
var=$1; vals=$2; reps=${3:-3}; shift 3 2>/dev/null
for rep in $(seq $reps); do
  for v in $vals; do
    env $var=$v python bench.py --no-extra-legs --no-cpu-baseline --steps 200 --warmup 30 "$@" 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j.get('mfma_classes_warmup',{}); print('$var=$v', j['ms_per_step'], j['final_loss'], {k:c[k]['frac'] for k in c})"
  done
done
