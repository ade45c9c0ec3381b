#!/usr/bin/env python3
"""bench.py's trainer_fit leg on its own (Trainer(config, model).fit(DeviceBatchLoader) at the headline shape): what a timeline / kernel
trace of the drop-in surface is taken from.  usage: python tools/fit_leg.py [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    sys.argv = sys.argv[:1]
    a = bench.parse()
    torch.cuda.set_device(0)
    print(json.dumps({"trainer_fit": bench.trainer_fit_leg(a, torch.device("cuda", 0), steps)}))
