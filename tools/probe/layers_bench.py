#!/usr/bin/env python3
"""Step time of the headline shape at other depths (the side stream serves up to 6 layers).  usage (GPU box): python tools/probe/layers_bench.py"""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for layers in (1, 2, 3, 4):
    for side in ("1", "0"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-gather-bench", "--no-extra-legs", "--n-items", "4000000",
                            "--layers", str(layers), "--steps", "100", "--warmup", "20"], capture_output=True, text=True, env=dict(os.environ, UR_SASREC_SIDE=side))
        j = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"layers {layers} side {side}: {j['ms_per_step']} ms/step loss {j['final_loss']}")
