#!/usr/bin/env python3
"""Steady-state timeline of a proved step from a rocprofv3 --kernel-trace CSV: per kernel (in launch order) the median duration and the median
idle gap in front of it.  python tools/r06_timeline.py <kernel_trace.csv> [marker substring of the step's dominant kernel = F16Unit] [kernels before it = 1]"""
import csv, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "F16Unit"
off = int(sys.argv[3]) if len(sys.argv) > 3 else 1
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
per = idx[-1] - idx[-2]
n_steps = min(30, len(idx) - 2)
start = idx[-1 - n_steps] - off
last = rows[start: start + per * n_steps]
steps = [last[i:i + per] for i in range(0, len(last), per)]
tot = 0
for j in range(per):
    dur = statistics.median((int(s[j]["End_Timestamp"]) - int(s[j]["Start_Timestamp"])) / 1e3 for s in steps)
    gaps = []
    for i, s in enumerate(steps):
        prev = s[j - 1] if j else (steps[i - 1][-1] if i else None)
        if prev is not None:
            gaps.append((int(s[j]["Start_Timestamp"]) - int(prev["End_Timestamp"])) / 1e3)
    gap = statistics.median(gaps)
    print(f"  {steps[0][j]['Kernel_Name'].split('(')[0][-46:]:46s} gap {gap:6.1f} us   kernel {dur:7.1f} us")
    tot += dur + gap
print(f"  {per} kernels per step; step (sum of medians): {tot:.1f} us")
