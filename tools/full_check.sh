#!/bin/bash
# full GPU validation of the tree: the gpu test suite, smoke(), the default bench line (outputs under gpurun_out/)
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/full_tests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 > gpurun_out/full_smoke.txt
timeout 900 python bench.py 2>gpurun_out/full_bench.err | tail -1 > gpurun_out/full_bench.json
cat gpurun_out/full_tests.txt gpurun_out/full_smoke.txt
python - <<'P'
import json
j=json.load(open('gpurun_out/full_bench.json'))
print(j['value'], j['ms_per_step'], j['roofline'])
print({k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if not isinstance(vv,(dict,list))}) for k,v in j.items() if k in ('e2e','dropout_0p5','zipf_ids','cpu_baseline')})
P
