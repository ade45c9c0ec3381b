#!/bin/bash
# usage (GPU box): bash tools/r6_lb.sh   -> W = 8 loopback ms per rank step under arithmetic switches
cd $GRAFT_REPO_ROOT
P='import sys,json; j=json.loads(sys.stdin.read())["loopback"]; print(sys.argv[1], j["ms_per_rank_step"], j["ms_per_step_wall_all_ranks_on_one_gpu"])'
python bench.py --loopback 8 --steps 10 --warmup 5 2>/dev/null | tail -1 | python -c "$P" default
UR_TEST=chain_split=0 python bench.py --loopback 8 --steps 10 --warmup 5 2>/dev/null | tail -1 | python -c "$P" chain_split=0
UR_TEST=nt_split=0 python bench.py --loopback 8 --steps 10 --warmup 5 2>/dev/null | tail -1 | python -c "$P" nt_split=0
UR_MFMA_ARITH=0 python bench.py --loopback 8 --steps 10 --warmup 5 2>/dev/null | tail -1 | python -c "$P" arith=0
python bench.py --loopback 8 --steps 10 --warmup 5 2>/dev/null | tail -1 | python -c "$P" default_again
