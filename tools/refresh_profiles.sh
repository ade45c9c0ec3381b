#!/bin/bash
# usage (on the GPU box): tools/refresh_profiles.sh <tag>  -> gpurun_out/<tag>_bench.json, _bench_kernel_stats.csv, _other_configs.txt, _gru_c4.json
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py 2>$out/${tag}_bench.err | grep '^{"metric"' | tail -1 > $out/${tag}_bench.json
python bench.py --steps 20000 --no-cpu-baseline --no-gather-bench --no-prof 2>/dev/null | grep '^{"metric"' | tail -1 > $out/${tag}_bench_long.json
bash tools/other_configs.sh > $out/${tag}_other_configs.txt 2>&1
python tools/gru_bench.py 2>/dev/null | grep '^{' | tail -1 > $out/${tag}_gru_c4.json
python tools/e2e_bench.py 2>/dev/null | tail -2 > $out/${tag}_e2e.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_r
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench > /tmp/prof_r.out 2>&1
cp /tmp/prof_r/r_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
grep '^{"metric"' /tmp/prof_r.out | tail -1 > $out/${tag}_bench_under_rocprof.json
