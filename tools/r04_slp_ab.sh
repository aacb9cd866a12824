#!/bin/bash
# the f16 scoring TUs with / without -fno-slp-vectorize (production: without SLP) on ONE box, two passes:  bash tools/r04_slp_ab.sh
cd "$(dirname "$0")/.."
for pass in 1 2; do
  for tag in base slp1 slp3; do
    lib=rails_amd/_ab/librails_amd_$tag.so; [ "$tag" = base ] && lib=rails_amd/librails_amd.so
    for pr in f16x1 f16x3; do
      echo -n "[$tag] $pr "; RAILS_AMD_LIBRARY=$lib python tools/score_bench.py --variants 0 --workload amzn-books --batch 32 --precision $pr --rounds 5 --reps 3 2>&1 | grep variant
    done
  done
done
