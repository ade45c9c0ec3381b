cp unirec_amd/libunirec_amd.so /tmp/tree.so
for rep in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then cp unirec_amd/libunirec_amd.so.base unirec_amd/libunirec_amd.so; else cp /tmp/tree.so unirec_amd/libunirec_amd.so; fi
    python bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=j['gather_roofline']['fused_gather_dot']; print('$v', f['frac'], f['frac_median'], j['gather_roofline']['frac'])"
  done
done
cp /tmp/tree.so unirec_amd/libunirec_amd.so
