#!/bin/bash
# PMC passes (separate runs) of one scoring kernel variant: tools/r04_pmc_small.sh <outdir> <variant> "<score_bench args>"
set -u
OUT=$1; VAR=$2; EXTRA=$3
export TMPDIR=/tmp
mkdir -p "$OUT"
CMD="python tools/score_bench.py --variants $VAR --rounds 1 --reps 2 $EXTRA"
i=0
for P in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA" \
         "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
         "FETCH_SIZE" "WRITE_SIZE" \
         "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
         "SQ_LEVEL_WAVES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-include-regex "mol_score" --output-format csv -d "$OUT/p$i" -o pmc -- $CMD > "$OUT/p$i.log" 2>&1 || echo "pass $i failed: $P"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k[0]:60s} {k[1]:32s} n={len(v):3d} mean/dispatch={sum(v)/len(v):.6g}")
PY
