#!/bin/bash
# the fused coarse top-K' alone on a 125 M x 32 bf16 table for each library under rails_amd/_ab plus the default build, twice
# (box drift), then the default build once under rocprofv3 for the per-kernel split:  bash tools/r04_c5_ab.sh <out tag> [batches]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/c5ab_$1; mkdir -p $O
BATCH=${2:-32,128}
for pass in 1 2; do
  for lib in default $(ls rails_amd/_ab/ 2>/dev/null); do
    [ $lib = default ] && unset RAILS_AMD_LIBRARY || export RAILS_AMD_LIBRARY=$PWD/rails_amd/_ab/$lib
    echo "[$lib] $(timeout 600 python tools/coarse_topk_bench.py --batch $BATCH --reps 20 2>&1 | grep 'N=' | tr '\n' '|')"
  done
done | tee $O/ab.txt
unset RAILS_AMD_LIBRARY
timeout 600 python tools/coarse_topk_bench.py --items 3000000 --check 3000000 --batch 32 --reps 5 2>&1 | grep -i "check\|N=" | tee $O/check.txt
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python tools/coarse_topk_bench.py --batch 32 --reps 20 > $O/prof.log 2>&1
python tools/kernel_stats_top.py $(find $O/prof -name '*kernel_stats.csv' | head -1) 14 2>/dev/null | tee $O/kernel_split.txt
