#!/bin/bash
# usage (GPU box): bash tools/r6_abl.sh <variant.so>   -> headline ms per step with the shipped library and with a variant build swapped in, two rounds
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-gather-bench --no-extra-legs --no-prof"
cp unirec_amd/libunirec_amd.so /tmp/lib_orig.so
for r in 1 2; do
  cp /tmp/lib_orig.so unirec_amd/libunirec_amd.so; echo -n "shipped  "; $B 2>&1 | grep -o '"ms_per_step": [0-9.]*' | tail -1
  cp $1 unirec_amd/libunirec_amd.so; echo -n "variant  "; $B 2>&1 | grep -o '"ms_per_step": [0-9.]*' | tail -1
done
cp /tmp/lib_orig.so unirec_amd/libunirec_amd.so
