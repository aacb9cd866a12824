#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 --kernel-trace CSV of tools/small_corpus_trace.py: the kernels of a step in
launch order with their median duration and the median idle gap in front of each (kernel end -> next kernel start).
  python tools/step_timeline.py <kernel_trace.csv> <kernels per step>"""
import csv, statistics, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
per = int(sys.argv[2])
rows = rows[-per * 150:]          # the last 150 steps (steady state)
assert len(rows) % per == 0
steps = [rows[i:i + per] for i in range(0, len(rows), per)]
names = [r["Kernel_Name"].split("(")[0][-44:] for r in steps[0]]
tot = []
for j in range(per):
    dur = statistics.median((int(s[j]["End_Timestamp"]) - int(s[j]["Start_Timestamp"])) / 1e3 for s in steps)
    gap = statistics.median((int(s[j]["Start_Timestamp"]) - int((s[j - 1] if j else steps[max(i - 1, 0)][-1])["End_Timestamp"])) / 1e3 for i, s in enumerate(steps) if j or i)
    print(f"{names[j]:46s} gap {gap:6.1f} us   kernel {dur:7.1f} us")
    tot.append(dur + gap)
print(f"step (sum of medians): {sum(tot):.1f} us")
