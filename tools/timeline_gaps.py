#!/usr/bin/env python3
"""GPU idle time inside the timed region of a bench.py run, from a rocprofv3 --kernel-trace CSV (tuning aid).
usage: timeline_gaps.py <kernel_trace.csv>   -> busy union over all queues, per-queue busy time, largest gaps and what follows them"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
# a window of steady-state steps: between two launches of the once-per-step plan kernel, well inside the timed region
import os
MARK = os.environ.get("MARK", "plan_chunk_sort_kernel")   # a kernel launched exactly once per step
marks = [e[0] for e in ev if MARK in e[2]]
n_steps = min(100, len(marks) - 12)
t0, t1 = marks[-(n_steps + 5)], marks[-5]
ev = [e for e in ev if t0 <= e[0] < t1]
print(f"{n_steps} steps: {(t1 - t0)/1e3/n_steps:.1f} us per step")
wall = ev[-1][1] - ev[0][0]
busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
gaps = []
for s, e, n, q in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"wall {wall/1e3:.0f} us, busy (union over queues) {busy/1e3:.0f} us = {100.0*busy/wall:.1f} %, kernels {len(ev)}")
perq = collections.Counter()
for s, e, n, q in ev: perq[q] += e - s
for q, t in perq.most_common(): print(f"  queue {q}: {t/1e3:.0f} us of kernels = {100.0*t/wall:.1f} % of wall")
after = collections.defaultdict(lambda: [0, 0])
for g, n in gaps:
    after[n[:60]][0] += g; after[n[:60]][1] += 1
print("idle time by the kernel that FOLLOWS the gap:")
for n, (g, c) in sorted(after.items(), key=lambda kv: -kv[1][0])[:12]: print(f"  {g/1e3:8.0f} us in {c:5d} gaps (avg {g/c/1e3:5.1f} us)  before {n}")
# gaps INSIDE the busiest queue (the main stream): time between the end of one of its kernels and the start of the next
mainq = perq.most_common(1)[0][0]
mq = [e for e in ev if e[3] == mainq]
qg = collections.defaultdict(lambda: [0, 0])
tot = 0
for (s0, e0, n0, _), (s1, e1, n1, _) in zip(mq, mq[1:]):
    g = max(0, s1 - e0)
    tot += g
    qg[(n0[:44], n1[:44])][0] += g; qg[(n0[:44], n1[:44])][1] += 1
print(f"main queue {mainq}: {tot/1e3/n_steps:.1f} us of gaps per step between its own kernels; largest, by (previous -> next):")
for (a, b), (g, c) in sorted(qg.items(), key=lambda kv: -kv[1][0])[:16]: print(f"  {g/1e3/n_steps:6.1f} us/step  avg {g/c/1e3:5.1f} us x{c/n_steps:4.1f}  {a}  ->  {b}")
print("per step, by queue and kernel:")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in ev:
    a = agg[(q, n[:72])]; a[0] += e - s; a[1] += 1
for (q, n), (t, c) in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1][0])):
    print(f"  q{q} {t/1e3/n_steps:7.1f} us  x{c/n_steps:4.1f}  {n}")
import os
if os.environ.get("SHOW"):
    key = os.environ["SHOW"]
    seqs = [(s, e - s, n) for s, e, n, q in ev if key in n]
    print(f"durations of the first 15 launches of *{key}* in the window (us):", [round(d / 1e3, 1) for _, d, _ in seqs[:15]])

if os.environ.get("DUMP"):   # one steady-state step, launch by launch: start offset, duration, queue, kernel (the critical path by eye)
    k = int(os.environ["DUMP"])
    a, b = marks[-(k + 6)], marks[-(k + 5)]
    print(f"one step ({(b - a)/1e3:.1f} us):")
    for s, e, n, q in ev:
        if a <= s < b:
            print(f"  {(s - a)/1e3:7.1f} +{(e - s)/1e3:6.1f}  q{q}  {n[:100]}")
