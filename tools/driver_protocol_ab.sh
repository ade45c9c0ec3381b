#!/bin/bash
# the driver's protocol (20 steps after 5 warm-ups, fresh process each) under two settings of one switch: tools/driver_protocol_ab.sh VAR "v1 v2" reps
var=$1; vals=$2; reps=${3:-4}
for rep in $(seq $reps); do for v in $vals; do
  env $var=$v python bench.py --gpus 1 --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', j['ms_per_step'], j['final_loss'])"
done; done
