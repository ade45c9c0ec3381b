#!/bin/bash
# usage: bash tools/kregs.sh unirec_amd/csrc/rowchain.hip [name filter]  -> registers / LDS / scratch per kernel of one source (compile only, no GPU)
src=$(realpath "$1"); filt=${2:-.}
tmp=$(mktemp -d); cd "$tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -c "$src" -save-temps -o k.o 2>/dev/null
python3 - "$filt" <<'PY'
import re, sys, glob
f = glob.glob('*gfx950.s')[0]
txt = open(f).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    if not re.search(sys.argv[1], name): continue
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\S+)', body) or [None, '?'])[1]
    import subprocess
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()[:70]
    print(f"{dn:70s} vgpr {g('next_free_vgpr'):>4s} agpr_off {g('accum_offset'):>4s} sgpr {g('next_free_sgpr'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s}")
PY
rm -rf "$tmp"
