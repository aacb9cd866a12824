#!/bin/bash
# Round 4: the small-unit scoring shell (RAILS_SCORE_VARIANT=7) against the dispatcher's choice (0), same process, interleaved rounds.
# Usage (on the GPU box): bash tools/r04_small_sweep.sh > gpurun_out/r04_small_sweep.txt
cd "$(dirname "$0")/.."
for wl in ml-1m ml-20m; do
  for b in 1 8 32; do
    echo "== $wl B=$b"; python tools/score_bench.py --variants 0,7 --workload $wl --batch $b --rounds 9 --reps 20 2>&1 | grep variant
  done
done
for b in 1 2 4 8 16 32; do
  echo "== amzn-books B=$b"; python tools/score_bench.py --variants 0,7 --workload amzn-books --batch $b --rounds 7 --reps 5 2>&1 | grep variant
done
for n in 86971 173941; do
  echo "== amzn-books shard N=$n B=32"; python tools/score_bench.py --variants 0,7 --workload amzn-books --items $n --batch 32 --rounds 9 --reps 10 2>&1 | grep variant
done
