#!/bin/bash
cd $GRAFT_REPO_ROOT
UR_TEST=tn_split=6 timeout 600 python -m pytest tests/test_gemm_gpu.py -k gemm_tn -q -x 2>&1 | tail -2
python tools/tn_group_bench.py 6 40 2>&1 | tail -1
UR_TEST=tn_split_target=256 python tools/tn_group_bench.py 6 40 2>&1 | tail -1
UR_TEST=tn_split_trace=2 python tools/tn_group_bench.py 6 8 2>&1 | tail -16
UR_TEST=tn_split_trace=2,tn_split_target=256 python tools/tn_group_bench.py 6 8 2>&1 | tail -16
python tools/tn_group_bench.py 0 40 2>&1 | tail -1
