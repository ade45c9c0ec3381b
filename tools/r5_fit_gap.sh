#!/bin/bash
# where Trainer.fit's step differs from the optimizer-level headline step: timelines of both on one box
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
{
echo "== headline (bench.py step loop)"
TAILN=70 bash tools/timeline.sh
echo "== Trainer.fit + DeviceBatchLoader (tools/fit_leg.py)"
CMD="python $GRAFT_REPO_ROOT/tools/fit_leg.py 200" MARK=plan_chunk_sort_kernel TAILN=80 bash tools/timeline.sh
} > gpurun_out/r5_fit_gap.txt 2>&1
cat gpurun_out/r5_fit_gap.txt
