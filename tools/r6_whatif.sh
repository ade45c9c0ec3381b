#!/bin/bash
# usage (GPU box): bash tools/r6_whatif.sh   -> chain kernel durations of the what-if builds (libunirec_amd_w<n>.so: -DRC_WHATIF=n), side stream off
cd $GRAFT_REPO_ROOT
export UR_SASREC_SIDE=0
cp unirec_amd/libunirec_amd.so /tmp/lib_orig.so
for n in 0 1 2 3 4 5 6; do
  [ $n = 0 ] && cp /tmp/lib_orig.so unirec_amd/libunirec_amd.so
  [ $n != 0 ] && { [ -f unirec_amd/libunirec_amd_w$n.so ] || continue; cp unirec_amd/libunirec_amd_w$n.so unirec_amd/libunirec_amd.so; }
  echo "== variant $n"; TAILN=40 bash tools/timeline.sh 2>&1 | grep "chain_"
done
cp /tmp/lib_orig.so unirec_amd/libunirec_amd.so
