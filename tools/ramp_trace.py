#!/usr/bin/env python3
"""First steps after a synchronisation, from a rocprofv3 --kernel-trace CSV of `bench.py --steps 20 --warmup 5` (tuning aid).
Per step of the LAST timed region (steps delimited by the once-per-step main-stream kernel MARK): wall to the next step, main-queue kernel
time, main-queue gap time; then the main-queue kernels of the first step next to the same kernels of a late step."""
import csv, sys, os, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows), key=lambda e: e[0])
MARK = os.environ.get("MARK", "compact_plan_kernel")
N = int(os.environ.get("NSTEPS", "20"))
marks = [e[0] for e in ev if MARK in e[2]]
perq = collections.Counter()
for s, e, n, q in ev: perq[q] += e - s
mainq = perq.most_common(1)[0][0]
marks = marks[-N:]
last_end = max(e[1] for e in ev)
bounds = marks + [last_end]
steps = []
for i in range(N):
    a, b = bounds[i], bounds[i + 1]
    mq = [e for e in ev if e[3] == mainq and a <= e[0] < b]
    ksum = sum(e[1] - e[0] for e in mq)
    gap = sum(max(0, y[0] - x[1]) for x, y in zip(mq, mq[1:]))
    steps.append((b - a, ksum, gap, mq))
    print(f"step {i:2d}: wall {(b-a)/1e3:7.1f} us  main-queue kernels {ksum/1e3:7.1f}  gaps {gap/1e3:6.1f}  launches {len(mq)}")
first, late = steps[0][3], steps[N // 2][3]
print("first step vs step", N // 2, "(main queue): offset, gap before, duration | late duration")
for k, (x, y) in enumerate(zip(first, late)):
    gb = x[0] - first[k - 1][1] if k else 0
    gl = y[0] - late[k - 1][1] if k else 0
    print(f"  {(x[0]-first[0][0])/1e3:7.1f}  gap {gb/1e3:6.1f} (late {gl/1e3:5.1f})  dur {(x[1]-x[0])/1e3:6.1f} | {(y[1]-y[0])/1e3:6.1f}  {x[2][:60]}")
# per kernel (all queues): average duration over the first 3 timed steps vs steps N-5..N-2
def agg(lo, hi):
    a, b = bounds[lo], bounds[hi]
    d = collections.defaultdict(float)
    for s, e, n, q in ev:
        if a <= s < b: d[(q, n[:70])] += (e - s) / (hi - lo)
    return d
A, B = agg(0, 3), agg(N - 5, N - 2)
print("kernel time per step, first 3 steps vs late steps (us), sorted by difference:")
for k in sorted(set(A) | set(B), key=lambda k: -(A.get(k, 0) - B.get(k, 0)))[:14]:
    print(f"  {A.get(k,0)/1e3:7.1f} {B.get(k,0)/1e3:7.1f}  {(A.get(k,0)-B.get(k,0))/1e3:+6.1f}  q{k[0]} {k[1]}")
# idle device between the last warm-up step and the first timed step
allm = [e[0] for e in ev if MARK in e[2]]
i0 = len(allm) - N
if i0 >= 1:
    prev_end = max(e[1] for e in ev if e[0] < allm[i0])
    print(f"idle between the last kernel before the timed region and its first kernel: {(allm[i0] - prev_end)/1e3:.1f} us")
    starts = [allm[k + 1] - allm[k] for k in range(max(0, i0 - 5), i0)]
    print("warm-up steps (start to start, us):", [round(x / 1e3, 1) for x in starts])
if i0 >= 1:
    a_, b_ = allm[i0 - 1], allm[i0]
    win = [e for e in ev if a_ <= e[0] < b_]
    t_end_step = a_ + 800_000
    late = [e for e in win if e[0] > t_end_step]
    agg2 = collections.Counter(); dur2 = collections.Counter()
    for s_, e_, n_, q_ in late: agg2[n_[:70]] += 1; dur2[n_[:70]] += e_ - s_
    print(f"between the last warm-up step and the timed region: {len(late)} launches, {sum(dur2.values())/1e3:.0f} us of kernels in {(b_ - t_end_step)/1e3:.0f} us")
    for n_, c_ in agg2.most_common(8): print(f"   x{c_:4d} {dur2[n_]/1e3:9.1f} us  {n_}")
    if late:
        gaps_ = sorted(((y[0] - x[1]) / 1e3 for x, y in zip(late, late[1:])), reverse=True)[:5]
        print("   largest idle gaps in that window (us):", [round(g, 1) for g in gaps_], " first launch", round((late[0][0] - t_end_step) / 1e3, 1), "us after the step")
