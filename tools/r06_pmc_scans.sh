#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (their own passes; x 2 for wide streaming reads on gfx950, MI355X_MICROARCH.md) of the round-6 streaming kernels:
# the component scans of MoLNaiveTopK5 (sample + select over the 356 MB table) and the candidate selection of the proved step (two passes over the 89 MB first-pass matrix)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r06pmc; mkdir -p $O
cd /tmp
for P in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $P --kernel-include-regex "coarse_scan_kernel" --output-format csv -d $O/naive_$P -o pmc -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms MoLNaiveTopK5 > /dev/null 2>&1
  timeout 400 rocprofv3 --pmc $P --kernel-include-regex "cand_hist|cand_compact|cand_finish" --output-format csv -d $O/cand_$P -o pmc -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 20 > /dev/null 2>&1
done
cd /root/repo
python - $O <<'PY' > $O/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("kernel | counter | dispatches | median per dispatch (the counter's unit: KB; FETCH_SIZE x 2 for wide streaming reads on gfx950)")
for k in sorted(agg):
    v = sorted(agg[k])
    print(f"{k[0]:62s} {k[1]:11s} n={len(v):3d} median={v[len(v)//2]:.6g}")
PY
rm -rf $O/naive_* $O/cand_*
