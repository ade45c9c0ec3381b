#!/bin/bash
mkdir -p gpurun_out
for g in 0 1; do UR_TN_GROUP=$g timeout 900 python bench.py --no-cpu-baseline --no-gather-bench --no-extra-legs --all-configs --steps 100 --warmup 20 2>/dev/null | tail -1 > gpurun_out/s7_all_group$g.json; done
python - <<'P'
import json
for g in (0,1):
    j=json.loads(open(f'gpurun_out/s7_all_group{g}.json').read())
    print(g, j['ms_per_step'], {k:(v.get('ms_per_step'), v.get('dominant'), v.get('frac')) if isinstance(v,dict) else v for k,v in j.get('other_configs',{}).items()})
P
{
echo "== lds"; timeout 900 bash tools/ab_env.sh UR_TN_LDS_KB "32 52 68" 2
echo "== early reduce"; timeout 900 bash tools/ab_env.sh UR_SASREC_EARLY_REDUCE "0 1" 2
} > gpurun_out/s7_ab.txt 2>&1
awk '{print $1,$2,$3}' gpurun_out/s7_ab.txt
