#!/bin/bash
# where the proved flow starts to pay at small batches: B = 1..8 on ML-20M (27 278 items) and B = 1 on amzn-books, proved flow forced against dense fp32
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06_probe_p; mkdir -p $O
{ for B in 1 2 4 8 16; do
    echo "== ml-20m B=$B proved flow (min-batch 1)"; python tools/exact_step_profile.py --precisions proved,fp32 --steps 400 --batch $B --min-batch 1 --workload ml-20m
  done
  for N in 100000 200000 400000; do for B in 1 2; do
    echo "== amzn-books config, $N items, B=$B"; python tools/exact_step_profile.py --precisions proved,fp32 --steps 300 --batch $B --min-batch 1 --items $N
  done; done; } 2>&1 | grep -v amdgpu | cut -c1-260 > $O/out2.txt
