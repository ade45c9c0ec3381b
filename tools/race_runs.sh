#!/bin/bash
# tools/race_runs.sh N ENV=VAL ... : N runs of the 200-step bench, prints the distinct final losses with their counts (more than one = a race)
n=$1; shift
for i in $(seq $n); do env "$@" python bench.py --no-extra-legs --no-cpu-baseline --steps 200 --warmup 30 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['final_loss'])"; done | sort | uniq -c
