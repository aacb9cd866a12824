#!/bin/bash
# sharded GPU tests at HEAD, per-rank shard steps (global proof), and where the proved flow starts to pay (corpus size sweep)
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s16; mkdir -p $O
python -m pytest tests/test_sharded_gpu.py -x -q -m gpu > $O/sharded_tests.txt 2>&1; echo "sharded tests rc=$?"; tail -3 $O/sharded_tests.txt
: > $O/shard_steps.txt
for R in 2 4 8; do
  for P in fp32 proved-global f16x3-exact; do
    python tools/shard_step_profile.py --world $R --precision $P 2>&1 | grep -v amdgpu.ids >> $O/shard_steps.txt
  done
done
cat $O/shard_steps.txt
: > $O/crossover.txt
for W in ml-1m ml-20m; do
  python tools/exact_step_profile.py --workload $W --precisions fp32,proved,f16x3 --min-items 0 --steps 200 --width 211 2>&1 | grep -v amdgpu.ids >> $O/crossover.txt
done
for N in 16384 32768 65536 131072 262144; do
  python tools/exact_step_profile.py --items $N --precisions fp32,proved --min-items 0 --steps 100 2>&1 | grep -v amdgpu.ids >> $O/crossover.txt
done
for B in 3 4 8 16; do
  python tools/exact_step_profile.py --batch $B --precisions fp32,proved --steps 100 2>&1 | grep -v amdgpu.ids >> $O/crossover.txt
done
cat $O/crossover.txt
