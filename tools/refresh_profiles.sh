#!/bin/bash
# usage (GPU box): tools/refresh_profiles.sh <tag> -> gpurun_out/<tag>_bench.json (the default bench.py line), _bench_driver_protocol.json (20 steps
# after 5 warm-ups), _bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the SAME command), _pmc_hbm_traffic.json (separate --pmc FETCH_SIZE /
# WRITE_SIZE passes), _pmc_mfma.json, _timeline.txt (per-queue busy time and gaps), _bench_all_configs.json, _sharded_w1.txt
tag=$1
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd $GRAFT_REPO_ROOT
python bench.py 2>$out/${tag}_bench.err | grep '^{"metric"' | tail -1 > $out/${tag}_bench.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1 > $out/${tag}_bench_driver_protocol.json
python bench.py --all-configs --no-cpu-baseline --no-gather-bench --steps 50 --warmup 10 2>/dev/null | grep '^{"metric"' | tail -1 > $out/${tag}_bench_all_configs.json
bash tools/sharded_w1_bench.sh 3 > $out/${tag}_sharded_w1.txt 2>&1
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_r
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r -o r -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gather-bench --no-extra-legs > /tmp/prof_r.out 2>&1
cp /tmp/prof_r/r_kernel_stats.csv $out/${tag}_bench_kernel_stats.csv
grep '^{"metric"' /tmp/prof_r.out | tail -1 > $out/${tag}_bench_under_rocprof.json
bash $GRAFT_REPO_ROOT/tools/pmc_hbm.sh $tag > /tmp/pmc_hbm.out 2>&1; tail -14 /tmp/pmc_hbm.out
rm -rf /tmp/pmc_m
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_m -o m -- python $GRAFT_REPO_ROOT/tools/cpu_bound_check.py > /tmp/pmc_m.out 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/pmc_m/m_counter_collection.csv $out/${tag}_pmc_mfma.json | head -14
cd $GRAFT_REPO_ROOT
TAILN=100 bash tools/timeline.sh > $out/${tag}_timeline.txt 2>&1
head -8 $out/${tag}_timeline.txt
cat $out/${tag}_sharded_w1.txt
(UR_TEST=plan_multi python tools/plan_bench.py; python tools/plan_bench.py) 2>/dev/null | grep ur_rows_plan > $out/${tag}_plan_bench.txt
(python tools/probe/gru_host.py 768; UR_TEST=gru_no_step python tools/probe/gru_host.py 768; python tools/gru_bench.py --hidden 768 --steps 30 | tail -1; UR_TEST=gru_no_step python tools/gru_bench.py --hidden 768 --steps 30 | tail -1) 2>/dev/null | grep -v amdgpu.ids > $out/${tag}_gru_h768.txt
(for m in "" 0; do UR_PREFETCH_ROWS=${m:-1} python bench.py --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 200 --warmup 30 --sharded-w1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sharded_w1 UR_PREFETCH_ROWS=${m:-1}', j['ms_per_step'], j['final_loss'])"; done) >> $out/${tag}_sharded_w1.txt
cat $out/${tag}_plan_bench.txt $out/${tag}_gru_h768.txt
