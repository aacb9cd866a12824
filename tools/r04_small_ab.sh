#!/bin/bash
# A/B of builds of the small-unit kernel on ONE box: tools/r04_small_ab.sh "<tags>" [workload batch items]...
cd "$(dirname "$0")/.."
tags=$1; shift
run() {  # workload batch items
  for tag in $tags; do
    lib=rails_amd/_ab/librails_amd_$tag.so; [ "$tag" = base ] && lib=rails_amd/librails_amd.so
    echo -n "$1 B=$2 N=${3:-full} [$tag] "
    RAILS_AMD_LIBRARY=$lib python tools/score_bench.py --variants 7 --workload $1 --batch $2 ${3:+--items $3} --rounds 7 --reps ${REPS:-5} 2>&1 | grep variant
  done
}
run amzn-books 32 173941
run amzn-books 8
run amzn-books 1
REPS=20 run ml-20m 32
REPS=20 run ml-1m 32
