#!/bin/bash
# usage (GPU box): tools/pmc_groups.sh <tag> "<counters of pass 1>" "<counters of pass 2>" ...   -> gpurun_out/<tag>_pmc_<i>.json (per-kernel
# sums per launch, tools/pmc_summary.py) of the default training step (tools/cpu_bound_check.py); kernel trace only, one pass per group
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  rm -rf /tmp/pg_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pg_$i -o p -- python $GRAFT_REPO_ROOT/tools/cpu_bound_check.py > /tmp/pg_$i.out 2>&1 \
    && python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pg_$i/p_counter_collection.csv $out/${tag}_pmc_$i.json | grep -E "${PMC_FILTER:-gemm_nt|gemm_tn|attn_bwd|chain}" | cut -c1-330 | head -14 \
    || echo "pass '$grp' failed: $(tail -3 /tmp/pg_$i.out)"
  i=$((i+1))
done
