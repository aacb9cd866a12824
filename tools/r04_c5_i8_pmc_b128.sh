cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/c5i8_b128; mkdir -p $O
for P in FETCH_SIZE WRITE_SIZE "SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $P | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $P --kernel-include-regex "coarse_scan_i8_kernel" --output-format csv -d $O/pmc_$n -o pmc -- python tools/coarse_topk_bench.py --batch 128 --reps 3 --prefilter on > $O/pmc_$n.log 2>&1
done
python - $O <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg): print(f"{k:32s} n={len(agg[k]):3d} mean/dispatch={sum(agg[k])/len(agg[k]):.6g}")
PY
