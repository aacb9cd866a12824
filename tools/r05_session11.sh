#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/s11; mkdir -p $O
rocprofv3 --hip-trace --kernel-trace --output-format csv -d $O/prof -o p -- python bench.py --steps 10 --warmup 3 --no-other-workloads --no-cpu-baseline --no-hr-parity > $O/bench.json 2> $O/bench.err
ls $O/prof
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/s11/prof/*hip_api_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
print(len(rows), "hip api calls; columns", list(rows[0].keys()))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
big = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Function"], (int(r["Start_Timestamp"]) - t0) / 1e9) for r in rows]
big.sort(reverse=True)
for d, fn, t in big[:60]:
    print(f"{d / 1e6:9.2f} ms  {fn:40s} at {t:8.3f} s")
import json
d = json.load(open("gpurun_out/s11/bench.json"))
for p in d["matrix"]: print(p["precision"], p["batch"], p["k_prime"], round(p["ms_per_step"], 3), round(p["ms_per_step_stdev"], 3))
PY
rm -rf $O/prof
