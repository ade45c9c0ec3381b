#!/bin/bash
# A/B of two builds on one box: the tree's library against unirec_amd/libunirec_amd.so.base (copied before the change)   usage: ab_build.sh [reps]
reps=${1:-3}
cp unirec_amd/libunirec_amd.so /tmp/new.so
for rep in $(seq $reps); do
  for v in base new; do
    if [ $v = base ]; then cp unirec_amd/libunirec_amd.so.base unirec_amd/libunirec_amd.so; else cp /tmp/new.so unirec_amd/libunirec_amd.so; fi
    python bench.py --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 200 --warmup 30 "${@:2}" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', j['ms_per_step'], j['final_loss'])"
  done
done
cp /tmp/new.so unirec_amd/libunirec_amd.so
