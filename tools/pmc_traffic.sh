#!/bin/bash
# HBM traffic of the scoring kernel per launch (FETCH_SIZE / WRITE_SIZE in their own passes, MI355X_MICROARCH.md: FETCH_SIZE x 2 on gfx950).
# usage: tools/pmc_traffic.sh <outdir> <score_bench args...>   (on the GPU box, from the repo root)
set -u
OUT=$1; shift
export TMPDIR=/tmp
mkdir -p "$OUT"
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $P --kernel-include-regex "mol_score" --output-format csv -d "$OUT/$P" -o pmc -- python tools/score_bench.py --variants 0 --rounds 1 --reps 2 "$@" > "$OUT/$P.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k[0]:60s} {k[1]:12s} n={len(v):3d} mean/dispatch={sum(v)/len(v):.6g} KB")
PY
