#!/bin/bash
# round 6 evidence, second pass (GPU box, final tree) -> gpurun_out/r06_h_* (tools/refresh_profiles.sh: default line, driver protocol, all configs,
# kernel stats, PMC, timeline, sharded world 1), r06_i (supervised world-1 RCCL line), r06_j (W = 8 loopback), the exact-arithmetic driver line
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd $GRAFT_REPO_ROOT
( time python bench.py > $out/r06_h_default_stdout.txt 2> $out/r06_h_default_stderr.txt ) 2>&1 | tail -4
bash tools/refresh_profiles.sh r06_h > $out/r06_h_refresh.log 2>&1; tail -12 $out/r06_h_refresh.log
bash tools/gather_pmc.sh r06_h > $out/r06_h_gather_pmc.log 2>&1; tail -3 $out/r06_h_gather_pmc.log
UR_MFMA_ARITH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather-bench --no-extra-legs 2>/dev/null | tail -1 > $out/r06_h_bench_driver_protocol_exact_fp32.json
python -m pytest tests/test_bench_ladder.py -q -m gpu -k "one_rank_through_the_supervisor" -s 2>&1 | grep "supervised world-1\|passed\|failed" > $out/r06_i_supervised_w1.txt; cat $out/r06_i_supervised_w1.txt
python bench.py --loopback 8 --steps 10 --warmup 5 2>/dev/null | tail -1 > $out/r06_j_w8_loopback.json; cut -c1-400 $out/r06_j_w8_loopback.json
