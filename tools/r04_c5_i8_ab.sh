#!/bin/bash
# builds of the int8 select scan (tiles per trip, waves per SIMD) and grid caps on ONE box; then the PMC traffic of the default build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/c5i8; mkdir -p $O
for pass in 1 2; do
  for lib in default $(ls rails_amd/_ab/ 2>/dev/null); do
    [ $lib = default ] && unset RAILS_AMD_LIBRARY || export RAILS_AMD_LIBRARY=$PWD/rails_amd/_ab/$lib
    echo "[$lib] $(timeout 600 python tools/coarse_topk_bench.py --batch 32 --reps 20 --prefilter on 2>&1 | grep 'N=' | cut -c1-90)"
  done
done | tee $O/ab.txt
unset RAILS_AMD_LIBRARY
for g in 1024 3072 4096 8192; do echo "grid $g: $(RAILS_SCAN8_GRID=$g python tools/coarse_topk_bench.py --batch 32 --reps 20 --prefilter on 2>&1 | grep N= | cut -c1-90)"; done | tee -a $O/ab.txt
for P in FETCH_SIZE WRITE_SIZE "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $P | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $P --kernel-include-regex "coarse_scan_i8_kernel" --output-format csv -d $O/pmc_$n -o pmc -- python tools/coarse_topk_bench.py --batch 32 --reps 3 --prefilter on > $O/pmc_$n.log 2>&1
done
python - $O <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print(f"{k:32s} n={len(agg[k]):3d} mean/dispatch={sum(agg[k])/len(agg[k]):.6g}")
PY
