F="--no-cpu-baseline --no-gather-bench --no-prof"
P='import sys,json; j=json.loads(sys.stdin.read()); print(sys.argv[1], j["ms_per_step"], j["value"])'
python bench.py $F --n-items 60000 --d 64 2>/dev/null | python -c "$P" C2
python bench.py $F --n-items 2000000 --seq-len 200 --negatives 1000 --loss softmax --batch 128 2>/dev/null | python -c "$P" C3
python bench.py $F --batch 4096 2>/dev/null | python -c "$P" C5_B4096
python bench.py $F --ids zipf 2>/dev/null | python -c "$P" C5_zipf
python bench.py $F --table-mode rowwise 2>/dev/null | python -c "$P" C5_rowwise
python bench.py $F --autograd --no-prefetch 2>/dev/null | python -c "$P" C5_autograd_noprefetch
python bench.py $F --dropout 0.5 2>/dev/null | python -c "$P" C5_dropout0.5
