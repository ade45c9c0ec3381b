#!/usr/bin/env python3
"""Launch geometry of every kernel of a steady-state step against the chip's resident-workgroup slots, from a rocprofv3 --kernel-trace CSV:
grid, workgroup, LDS, VGPR+AGPR -> workgroups per CU (registers: 512 per SIMD lane in granules of 8; LDS 160 KB; 32 waves per CU... 8 per SIMD),
slots = 256 CUs x that, rounds = grid / slots.  A grid just above a multiple of the slots pays a whole extra round for its tail.
usage: occupancy.py <kernel_trace.csv>"""
import csv, sys, collections, math, os
rows = list(csv.DictReader(open(sys.argv[1])))
MARK = os.environ.get("MARK", "plan_chunk_sort_kernel")
marks = sorted(int(r["Start_Timestamp"]) for r in rows if MARK in r["Kernel_Name"])
t0, t1 = marks[-105], marks[-5]
agg = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s = int(r["Start_Timestamp"])
    if not (t0 <= s < t1): continue
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // wg
    key = (r["Kernel_Name"][:70], grid, wg, int(r["LDS_Block_Size"]), int(r["VGPR_Count"]), int(r.get("Accum_VGPR_Count", 0) or 0), int(r["Scratch_Size"]), r["Queue_Id"])
    a = agg.setdefault(key, [0, 0])
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - s
print(f"{'kernel':70s} {'q':>2s} {'grid':>6s} {'wg':>4s} {'LDS':>6s} {'regs':>4s} {'scr':>4s} {'wg/CU':>5s} {'slots':>5s} {'rounds':>6s} {'us':>7s} {'n/step':>6s}")
for (name, grid, wg, lds, v, ag, scr, q), (n, t) in agg.items():
    waves = (wg + 63) // 64
    regs = 2 * v + ag                                      # the trace's VGPR_Count is in units of 2 registers (84 = the compiler's 168); dynamic LDS is not in the trace (0)
    regs_al = (regs + 7) // 8 * 8
    wps = max(1, min(8, 512 // max(regs_al, 1)))           # waves per SIMD by registers
    by_regs = wps * 4 // waves if waves <= 4 else wps // ((waves + 3) // 4)
    by_lds = (160 * 1024) // lds if lds else 99
    by_waves = 32 // waves
    per_cu = max(1, min(by_regs, by_lds, by_waves))
    slots = 256 * per_cu
    print(f"{name:70s} {q:>2s} {grid:6d} {wg:4d} {lds:6d} {regs:4d}+{ag:<3d} {scr:4d} {per_cu:5d} {slots:5d} {grid/slots:6.2f} {t/n/1e3:7.1f} {n/100:6.2f}")
