#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 4 2; do
  echo "--- gru_rows=$r"
  UR_TEST=gru_rows=$r timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases_gpu.py tests/test_dropout_gpu.py -q -m gpu -k "gru or GRU" 2>&1 | tail -2
  UR_TEST=gru_rows=$r python tools/gru_bench.py --steps 50 2>&1 | tail -2
  UR_TEST=gru_rows=$r python tools/gru_bench.py --steps 50 --hidden 64 2>&1 | tail -1
done
