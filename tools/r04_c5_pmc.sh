#!/bin/bash
# PMC passes for the coarse select scan on a 125 M x 32 bf16 table (tools/coarse_topk_bench.py; separate --pmc runs): bash tools/r04_c5_pmc.sh <outdir> <batch>
set -u
OUT=${1:-gpurun_out/c5pmc}; B=${2:-128}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p "$OUT"
CMD="python tools/coarse_topk_bench.py --batch $B --reps 3"
P1="GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA"
P2="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_SCA"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $P --kernel-include-regex "coarse_scan_kernel<2, 2" --output-format csv -d "$OUT/p$i" -o pmc -- $CMD > "$OUT/p$i.log" 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:32s} n={len(v):3d} mean/dispatch={sum(v)/len(v):.6g}")
PY
