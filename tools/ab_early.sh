#!/bin/bash
# A/B of the lazy-Adam replay's placement (plan stream under the step vs tail of step()) over the headline legs and every other config, one box
for rep in 1 2; do for x in 1 0; do
UR_EARLY_CATCHUP=$x python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-gather-bench --all-configs 2>/dev/null | tail -1 > gpurun_out/ab_$x.json; python - <<PY
import json
j=json.load(open("gpurun_out/ab_$x.json"))
print("EARLY=$x headline", j["ms_per_step"], "e2e", j["e2e"]["ms_per_step"], "fit", j["trainer_fit"]["ms_per_step"], "steady", j["steady_state"]["ms_per_step"], "zipf", j.get("zipf_ids",{}).get("ms_per_step"), " | ", " ".join(f"{k}={v.get('ms_per_step')}" for k,v in j.get("other_configs",{}).items()), "C3e2e", j["other_configs"]["C3"].get("e2e",{}).get("ms_per_step"))
PY
done; done
