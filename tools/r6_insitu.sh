#!/bin/bash
# in-situ check of the split dW kernel: headline A/B + per-kernel durations
cd $GRAFT_REPO_ROOT
for m in "tn_split=0" "tn_split=6" "tn_split=6,tn_split_target=384" "tn_split=6,tn_split_target=768" "tn_split=0" "tn_split=6"; do
  UR_TEST=$m python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-bench 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$m headline ms', j['ms_per_step'], 'fit', j.get('trainer_fit',{}).get('ms_per_step'))"
done
bash tools/kstats.sh r06c_k6 UR_TEST=tn_split=6 -- 2>&1 | head -14
