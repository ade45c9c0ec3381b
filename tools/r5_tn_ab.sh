#!/bin/bash
# round 5: the 128 x 128 LDS-DMA weight-gradient kernel against the 64 x 64 one -- parity, isolated rates, the step in situ (interleaved)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gemm_gpu.py -q -m gpu -x -k "tn" 2>&1 | tail -4 > gpurun_out/r5_tn_tests.txt
cat gpurun_out/r5_tn_tests.txt
{
for big in 0 1; do
    echo "== UR_TN_BIG=$big"
    UR_TN_BIG=$big timeout 300 python tools/gemm_bench.py 21248 2>&1 | grep "^TN"
done
} > gpurun_out/r5_tn_isolated.txt 2>&1
cat gpurun_out/r5_tn_isolated.txt
{
for rep in 1 2 3; do
  for spec in "0 0 1" "1 0 1" "1 192 1" "1 384 1" "0 0 0" "1 0 0"; do
    set -- $spec
    UR_TN_BIG=$1 UR_TN_TARGET=$2 UR_SASREC_SIDE=$3 timeout 600 python bench.py --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 200 --warmup 30 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=j.get('mfma_classes_warmup',{}); print('BIG=$1 TARGET=$2 SIDE=$3', j['ms_per_step'], j['final_loss'], {k:c[k]['frac'] for k in c})"
  done
done
} > gpurun_out/r5_tn_insitu.txt 2>&1
cat gpurun_out/r5_tn_insitu.txt
