#!/bin/bash
# usage (GPU box): tools/loopback_bench.sh <tag>  -> gpurun_out/<tag>_w8_loopback.txt: the multi-GPU step at true W-rank shapes on ONE GPU
# (W rank threads through the in-process loopback transport) next to the single-GPU step and the sharded path at world 1, same box
tag=${1:-r05}
out=gpurun_out/${tag}_w8_loopback.txt; mkdir -p gpurun_out
{
echo "# single-GPU step (plain optimizer) and the sharded path at world 1, 200 steps after 30 warm-ups:"
python bench.py --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 200 --warmup 30 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('single_gpu ms_per_step', j['ms_per_step'], 'loss', j['final_loss'])"
python bench.py --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 200 --warmup 30 --sharded-w1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sharded_w1 ms_per_step', j['ms_per_step'], 'loss', j['final_loss'])"
for W in 2 4 8; do
  echo "# --loopback $W (100 steps after 20 warm-ups):"
  timeout 1200 python bench.py --loopback $W --steps 100 --warmup 20 2>gpurun_out/${tag}_loopback_$W.err | tail -1
  UR_PREFETCH_ROWS=0 timeout 1200 python bench.py --loopback $W --steps 100 --warmup 20 2>/dev/null | tail -1 | sed 's/^/UR_PREFETCH_ROWS=0 /'
done
} > $out 2>&1
cat $out
