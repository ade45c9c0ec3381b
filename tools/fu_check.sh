python -m pytest tests/test_gpu_parity.py -q -m gpu -k "reduce_update" 2>&1 | tail -1
TAILN=26 bash tools/timeline.sh 2>&1 | grep -E "ms_per_step|rows_reduce|sparse_adam|us per step|gemm_tn|reduce_batch"
for i in 1 2 3; do python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs 2>/dev/null | tail -1 | cut -c40-130; done
python bench.py --n-items 2000000 --seq-len 200 --negatives 1000 --loss softmax --batch 128 --no-cpu-baseline --no-gather-bench --no-prof --no-extra-legs --steps 100 --warmup 20 2>/dev/null | tail -1 | cut -c40-140
