#!/usr/bin/env python3
"""Per (shape, batch, mode) GPU time of the query prologue from a rocprofv3 kernel trace of tools/r04_prologue_ab.py:
  rocprofv3 --kernel-trace -d out -o prol --output-format csv -- python tools/r04_prologue_ab.py; python tools/r04_prologue_stats.py out/prol_kernel_trace.csv
The tool calls query_pack 51 x 9 times per (shape, batch, mode) in a fixed order; kernels are attributed by launch order."""
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "query_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per_call = {"1": 1, "2": 3, "3": 2}
i = 0
for name in ("ml-1m", "ml-20m", "amzn-books"):
    for B in (1, 8, 32, 128):
        acc = collections.defaultdict(list)
        for rnd in range(9):
            for m in ("1", "2", "3"):
                for call in range(51):
                    ks = rows[i : i + per_call[m]]
                    i += per_call[m]
                    span = (int(ks[-1]["End_Timestamp"]) - int(ks[0]["Start_Timestamp"])) / 1e3
                    busy = sum(int(k["End_Timestamp"]) - int(k["Start_Timestamp"]) for k in ks) / 1e3
                    acc[m].append((span, busy))
        med = lambda v: sorted(v)[len(v) // 2]
        print(f"{name:10s} B={B:3d}  " + "   ".join(f"{lbl} span {med([s for s, _ in acc[m]]):5.1f} busy {med([b for _, b in acc[m]]):5.1f} us" for m, lbl in (("1", "per-query"), ("2", "batched"), ("3", "split"))))
assert i == len(rows), (i, len(rows))
