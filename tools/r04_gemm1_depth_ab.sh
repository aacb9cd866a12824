#!/bin/bash
# A/B of GEMM1 lookahead depth builds (RAILS_GEMM1_DEPTH, RAILS_STAGED1_PIPE) on ONE box, two passes so box drift shows
cd "$(dirname "$0")/.."
tags=${1:-"base d2 d3 s1p s1d3"}
run() {  # variant workload batch items
  for tag in $tags; do
    lib=rails_amd/_ab/librails_amd_$tag.so; [ "$tag" = base ] && lib=rails_amd/librails_amd.so
    echo -n "v$1 $2 B=$3 N=${4:-full} [$tag] "
    RAILS_AMD_LIBRARY=$lib python tools/score_bench.py --variants $1 --workload $2 --batch $3 ${4:+--items $4} --rounds 7 --reps ${REPS:-20} 2>&1 | grep variant
  done
}
for pass in 1 2; do
  run 0 ml-20m 8
  run 0 ml-20m 32
  run 0 ml-20m 64 221184
  run 0 ml-1m 32
done
REPS=3 run 0 amzn-books 32 695764
