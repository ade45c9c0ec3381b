#!/bin/bash
# usage (GPU box): tools/pmc_hbm.sh <tag>  -> gpurun_out/<tag>_pmc_hbm_traffic.json   (two --pmc passes, no other tracing domains)
tag=$1; out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pmc_f /tmp/pmc_w
CMD="python $GRAFT_REPO_ROOT/tools/cpu_bound_check.py"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- $CMD > /tmp/pmc_f.out 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- $CMD > /tmp/pmc_w.out 2>&1
ls /tmp/pmc_f /tmp/pmc_w
python $GRAFT_REPO_ROOT/tools/pmc_hbm.py /tmp/pmc_f/f_counter_collection.csv /tmp/pmc_w/w_counter_collection.csv $out/${tag}_pmc_hbm_traffic.json "python tools/cpu_bound_check.py (60 steps of the bench.py workload)"
