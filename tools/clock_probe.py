#!/usr/bin/env python3
"""Engine / memory / fabric clock levels of THIS process's GPU (sysfs, matched by PCI bus id), sampled by a background thread every ~1 ms
while the first training steps of the process run: is the slow start (first ~20 steps 3-7 % slower) a clock ramp?"""
import os, sys, time, gc, glob, threading
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

props = torch.cuda.get_device_properties(0)
bus = getattr(props, "pci_bus_id", None)
dom = getattr(props, "pci_domain_id", 0)
dev_id = getattr(props, "pci_device_id", 0)
want = f"{dom:04x}:{bus:02x}:{dev_id:02x}" if bus is not None else None
card = None
for d in glob.glob("/sys/class/drm/card*/device"):
    real = os.path.realpath(d)
    if want and want in real and os.path.exists(d + "/pp_dpm_sclk"):
        card = d
print("device", props.name, "pci", want, "->", card)
if card is None:
    cards = [d for d in glob.glob("/sys/class/drm/card*/device") if os.path.exists(d + "/pp_dpm_sclk")]
    print("no match; candidates:", [os.path.realpath(c)[-12:] for c in cards])
    sys.exit(0)
samples = []
stop = False
def poll():
    files = [card + "/pp_dpm_sclk", card + "/pp_dpm_mclk", card + "/pp_dpm_fclk", card + "/pp_dpm_socclk"]
    busy = card + "/gpu_busy_percent"
    while not stop:
        t = time.perf_counter()
        row = [cur(f) for f in files]
        try: row.append(open(busy).read().strip() + "%")
        except Exception: row.append("?")
        samples.append((t, row))
        time.sleep(0.001)
a = bench.parse()
dev = torch.device("cuda:0")
model = SASRec(bench.model_config(a, "cuda:0"))
opt = SparseDenseAdam(model, lr=1e-3, table_mode=a.table_mode)
model.train()
batches = bench.synth_batches(a, a.n_items, dev, 1, n_batches=130)
def step(b, nxt):
    opt.zero_grad()
    opt.plan_batch(item_seq=b["item_seq"], item_id=b["item_id"])
    opt.prefetch_plan(item_seq=nxt["item_seq"], item_id=nxt["item_id"])
    model.forward_backward(item_id=b["item_id"], label=b["label"], item_seq=b["item_seq"])
    opt.step(late_join=True)
gc.disable()
torch.cuda.synchronize(); time.sleep(0.3)
th = threading.Thread(target=poll, daemon=True); th.start()
time.sleep(0.01)
marks = []
k = 0
for chunk in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); t0 = time.perf_counter()
    for i in range(10):
        step(batches[k], batches[k + 1]); k += 1
    e1.record(); e1.synchronize(); t1 = time.perf_counter()
    marks.append((t0, t1, e0.elapsed_time(e1) / 10))
stop = True; th.join()
for (t0, t1, ms) in marks:
    rows = [r for (t, r) in samples if t0 <= t <= t1]
    uniq = {}
    for r in rows: uniq[tuple(r)] = uniq.get(tuple(r), 0) + 1
    print(f"{ms:.4f} ms/step  samples {len(rows):3d}  " + "  ".join(f"{'/'.join(k)} x{v}" for k, v in sorted(uniq.items(), key=lambda kv: -kv[1])[:3]))
pre = [r for (t, r) in samples if t < marks[0][0]]
print("idle before:", pre[-1] if pre else None)
