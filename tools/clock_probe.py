#!/usr/bin/env python3
"""Engine / memory / fabric clock levels (sysfs) sampled while training steps are queued: after how many steps does the device sit at its
sustained clocks?"""
import os, sys, time, gc, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from unirec_amd.facility.optimizer import SparseDenseAdam
from unirec_amd.model.sequential.sasrec import SASRec
def cur(path):
    try:
        for line in open(path).read().splitlines():
            if line.strip().endswith("*"): return line.split(":")[1].strip(" *")
    except Exception as e:
        return f"?({type(e).__name__})"
    return "?"
cards = [d for d in glob.glob("/sys/class/drm/card*/device") if os.path.exists(d + "/pp_dpm_sclk")]
print("cards:", cards)
def clocks():
    out = []
    for d in cards[:1]:
        out += [cur(d + "/pp_dpm_sclk"), cur(d + "/pp_dpm_mclk"), cur(d + "/pp_dpm_fclk")]
        for h in glob.glob(d + "/hwmon/hwmon*/power1_average") + glob.glob(d + "/hwmon/hwmon*/power1_input"):
            try: out.append(f"{int(open(h).read())/1e6:.0f}W")
            except Exception: pass
    return " ".join(out)
a = bench.parse()
dev = torch.device("cuda:0")
model = SASRec(bench.model_config(a, "cuda:0"))
opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
model.train()
batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=200)
def step(b, nxt):
    opt.zero_grad()
    opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
    opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
    model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
    opt.step()
gc.collect(); gc.disable()
torch.cuda.synchronize(); time.sleep(0.2)
print("idle:", clocks())
k = 0
for chunk in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(10):
        step(batches[k], batches[k + 1]); k += 1
    e1.record()
    c = clocks()          # read while the device is still working on the chunk (the host runs ahead)
    e1.synchronize()
    print(f"steps {k-10:3d}-{k-1:3d}: {e0.elapsed_time(e1)/10:.4f} ms/step  clocks while busy: {c}")
