#!/usr/bin/env python3
"""Per-launch-geometry averages of the HSTU encoder's kernels from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d DIR -o h -- python tools/hstu_bench.py ; python tools/hstu_kernel_split.py DIR"""
import collections
import csv
import glob
import sys

f = (glob.glob(sys.argv[1] + "/*/h_kernel_trace.csv") + glob.glob(sys.argv[1] + "/h_kernel_trace.csv"))[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if n.startswith("mol::") or "mol::" in n:
        d[(n.split("(")[0][-40:], r["Grid_Size_X"], r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:42s} grid {k[1]:>8s} x {k[2]:>4s} x {k[3]:>4s}  launches {len(v):5d}  avg {sum(v) / len(v):8.1f} us")
