#!/bin/bash
# round 6 evidence (GPU box, final tree) -> gpurun_out/r06_*: stage-A table of the split-bf16 arithmetic, workgroup traces, PMC of the split and
# the exact weight-gradient kernels, the MT19937 row builder's rows/s, the supervised world-1 RCCL line, then tools/refresh_profiles.sh r06_h
out=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $out
cd $GRAFT_REPO_ROOT
python tools/split_bf16_stage_a.py 21248 $out/r06_a_stage_a.json > $out/r06_a_stage_a.txt 2>&1; head -5 $out/r06_a_stage_a.txt
bash tools/r6_trace.sh > $out/r06_b_tn_split_trace.txt 2>&1; grep "first start" $out/r06_b_tn_split_trace.txt
(echo "=== split-bf16 (arith 6)"; ARITH=6 bash tools/tn_group_pmc.sh; echo "=== exact fp32 MFMA (arith 0)"; ARITH=0 bash tools/tn_group_pmc.sh) > $out/r06_c_tn_split_pmc.txt 2>&1; grep "us avg" $out/r06_c_tn_split_pmc.txt
python -m pytest tests/test_data_path.py -q -m gpu -k "mt_device_builder_equals_the_host_builder_on_1e5" -s 2>&1 | grep "MT19937 row builders\|passed\|failed" > $out/r06_g_mt_rows.txt; cat $out/r06_g_mt_rows.txt
python tools/input_bench.py 2>/dev/null | tail -3 >> $out/r06_g_mt_rows.txt
python -m pytest tests/test_bench_ladder.py -q -m gpu -k "one_rank_through_the_supervisor" -s 2>&1 | grep "supervised world-1\|passed\|failed" > $out/r06_i_supervised_w1.txt; cat $out/r06_i_supervised_w1.txt
python bench.py --loopback 8 --steps 10 --warmup 5 2>/dev/null | tail -1 > $out/r06_j_w8_loopback.json; cut -c1-400 $out/r06_j_w8_loopback.json
( time python bench.py > $out/r06_h_default_stdout.txt 2> $out/r06_h_default_stderr.txt ) 2>&1 | tail -4
bash tools/refresh_profiles.sh r06_h > $out/r06_h_refresh.log 2>&1; tail -12 $out/r06_h_refresh.log
bash tools/gather_pmc.sh r06_h > $out/r06_h_gather_pmc.log 2>&1; tail -3 $out/r06_h_gather_pmc.log
UR_MFMA_ARITH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gather-bench --no-extra-legs 2>/dev/null | tail -1 > $out/r06_h_bench_driver_protocol_exact_fp32.json
