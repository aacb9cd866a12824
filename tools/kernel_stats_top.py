#!/usr/bin/env python3
"""Top rows of a rocprofv3 --stats kernel_stats.csv:  python tools/kernel_stats_top.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 15]:
    avg = float(r["AverageNs"]) / 1e3
    print(f"{r['Name'][:110]:110s} {r['Calls']:>6s} {avg:10.1f} us  {float(r['Percentage']):6.2f} %")
