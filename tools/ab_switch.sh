#!/bin/bash
# usage: [SW=UR_EARLY_CATCHUP] bash tools/ab_switch.sh -- A/B of one 0/1 switch over the headline legs and every other config, one box
SW=${SW:-UR_FUSED_UPDATE}
for rep in 1 2; do for x in 1 0; do
env $SW=$x python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-bench --all-configs 2>/dev/null | tail -1 > gpurun_out/ab_$x.json; python - <<PY
import json
j=json.load(open("gpurun_out/ab_$x.json"))
print("$SW=$x headline", j["ms_per_step"], "e2e", j["e2e"]["ms_per_step"], "fit", j["trainer_fit"]["ms_per_step"], "steady", j["steady_state"]["ms_per_step"], "zipf", j.get("zipf_ids",{}).get("ms_per_step"), " | ", " ".join(f"{k}={v.get('ms_per_step')}" for k,v in j.get("other_configs",{}).items()), "C3e2e", j["other_configs"]["C3"].get("e2e",{}).get("ms_per_step"))
PY
done; done
