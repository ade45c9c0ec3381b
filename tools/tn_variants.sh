#!/bin/bash
# usage (GPU box): tools/tn_variants.sh -> the weight-gradient kernel variants (64 x 64 shipped; UR_TEST=tn_big=128 / 64: LDS-DMA 128 x 128 / 128 x 64) :
# parity of the d = 128-class shapes, the headline step, per-kernel durations in situ, MFMA busy + HBM bytes (separate --pmc passes)
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "" tn_big=128 tn_big=64; do
  echo "=== UR_TEST=$v"
  UR_TEST=$v timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -k "gemm_tn" 2>&1 | tail -1
  for i in 1 2; do UR_TEST=$v python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('  headline ms_per_step', j['ms_per_step'], 'ex/s', j['value'])"; done
  UR_TEST=$v bash tools/kstats.sh tnv -- --no-extra-legs 2>&1 | grep -E "gemm_tn|reduce_batch|rows_reduce|attn_bwd|chain_proj_bwd|sum of"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tnv_m /tmp/tnv_f /tmp/tnv_w
    UR_TEST=$v rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/tnv_m -o m -- python $GRAFT_REPO_ROOT/tools/cpu_bound_check.py > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/tnv_m/m_counter_collection.csv | grep -E "gemm_tn" | cut -c1-260
    UR_TEST=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/tnv_f -o m -- python $GRAFT_REPO_ROOT/tools/cpu_bound_check.py > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/tnv_f/m_counter_collection.csv | grep -E "gemm_tn" | cut -c1-200
    UR_TEST=$v rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/tnv_w -o m -- python $GRAFT_REPO_ROOT/tools/cpu_bound_check.py > /dev/null 2>&1
    python $GRAFT_REPO_ROOT/tools/pmc_summary.py /tmp/tnv_w/m_counter_collection.csv | grep -E "gemm_tn" | cut -c1-240 )
done
