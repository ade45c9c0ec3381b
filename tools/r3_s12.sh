#!/bin/bash
mkdir -p gpurun_out
timeout 1200 python bench.py --all-configs --steps 20 --warmup 5 > gpurun_out/s12_bench_all.json 2> gpurun_out/s12_bench_all.err
tail -3 gpurun_out/s12_bench_all.err
python - <<'P'
import json
j=json.loads(open('gpurun_out/s12_bench_all.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','roofline','e2e','dropout_0p5','zipf_ids','steady_state','other_configs','cpu_baseline'):
    print(k, json.dumps(j.get(k))[:900])
print('gather', json.dumps(j.get('gather_roofline'))[:1500])
P
