for cfg in "--n-items 60000 --d 64" "--n-items 2000000 --seq-len 200 --negatives 1000 --loss softmax --batch 128" "--n-items 1000000 --dropout 0.2" "--n-items 1000000 --layers 1" "--n-items 1000000 --layers 3"; do
  echo "== $cfg"
  for i in 1 2 3 4 5 6; do python bench.py $cfg --no-extra-legs --no-cpu-baseline --no-gather-bench --steps 100 --warmup 10 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['final_loss'])"; done | sort | uniq -c
done
