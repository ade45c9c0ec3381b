#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "tn_split_minwg=0" "tn_split_minwg=384" "tn_split_minwg=97" "tn_split_minwg=0" "tn_split_minwg=384" "tn_split_minwg=97"; do
  UR_TEST=$v python bench.py --all-configs --no-cpu-baseline --no-gather-bench --no-extra-legs --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('$v headline', j['ms_per_step'], ' '.join(k+'='+str(v['ms_per_step']) for k,v in j['other_configs'].items()))"
done
