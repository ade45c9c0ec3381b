#!/bin/bash
# A/B/C... of several builds and/or environments on one box, interleaved:
#   tools/ab_multi.sh <reps> "<name>:<so file>[:VAR=val[,VAR=val]]" ...      (so file relative to unirec_amd/, '-' = the tree's library)
reps=$1; shift
cp unirec_amd/libunirec_amd.so /tmp/tree.so
for rep in $(seq $reps); do
  for spec in "$@"; do
    IFS=: read name so envs <<< "$spec"
    if [ "$so" = "-" ]; then cp /tmp/tree.so unirec_amd/libunirec_amd.so; else cp unirec_amd/$so unirec_amd/libunirec_amd.so; fi
    env $(echo $envs | tr ',' ' ') python bench.py --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 200 --warmup 30 $AB_ARGS 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', j['ms_per_step'], j['final_loss'])"
  done
done
cp /tmp/tree.so unirec_amd/libunirec_amd.so
