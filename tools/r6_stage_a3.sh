#!/bin/bash
# round 6, stage A third pass (GPU box): split math under the MFMAs -- variants of the interleave hint / prefetch depth / workgroup target
cd $GRAFT_REPO_ROOT
UR_TEST=tn_split=6 timeout 600 python -m pytest tests/test_gemm_gpu.py -k gemm_tn -q -x 2>&1 | tail -2
for v in "tn_split_il=0" "tn_split_il=4" "tn_split_il=6" "tn_split_il=8" "tn_split_pf=2,tn_split_il=6" "tn_split_pf=2,tn_split_il=0" "tn_split_il=6,tn_split_target=256" "tn_split_il=6,tn_split_target=384" "tn_split_il=6,tn_split_target=640"; do
  UR_TEST=$v python tools/tn_group_bench.py 6 40 2>&1 | tail -1
done
python tools/tn_group_bench.py 0 40 2>&1 | tail -1
