#!/bin/bash
# usage (GPU box): bash tools/r6_abk.sh <variant.so> [more variants]  -> chain kernel durations (side stream off, then in situ) of the shipped library and of variant builds
cd $GRAFT_REPO_ROOT
cp unirec_amd/libunirec_amd.so /tmp/lib_orig.so
for v in shipped "$@"; do
  [ $v = shipped ] && cp /tmp/lib_orig.so unirec_amd/libunirec_amd.so || cp $v unirec_amd/libunirec_amd.so
  echo "== $v (alone)"; UR_SASREC_SIDE=0 TAILN=40 bash tools/timeline.sh 2>&1 | grep "chain_"
  echo "== $v (in situ)"; TAILN=40 bash tools/timeline.sh 2>&1 | grep "chain_\|ms_per_step"
done
cp /tmp/lib_orig.so unirec_amd/libunirec_amd.so
