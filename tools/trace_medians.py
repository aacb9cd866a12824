#!/usr/bin/env python3
"""Median kernel durations from a rocprofv3 --kernel-trace CSV, grouped by (kernel, grid, workgroup): pooled --stats
averages mix workloads and cold first calls.  usage: trace_medians.py <kernel_trace.csv> [substring ...]"""
import csv, statistics, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
d = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if want and not any(w in n for w in want):
        continue
    d[(n.split("(")[0][-48:], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", ""))].append(
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -statistics.median(kv[1]) * len(kv[1])):
    print(f"{k[0]:50s} grid {k[1]:>8s}x{k[2]:<3s} wg {k[3]:>5s}  calls {len(v):4d}  median {statistics.median(v):9.1f} us  min {min(v):9.1f}  max {max(v):9.1f}")
