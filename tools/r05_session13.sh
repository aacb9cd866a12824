#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s13; mkdir -p $O
timeout 1500 python -m pytest tests/test_sharded_gpu.py -q -x > $O/sharded_tests.log 2>&1; echo "sharded tests rc=$?"; tail -5 $O/sharded_tests.log
for R in 2 4 8; do
  for P in "" "--precision proved-global" "--precision f16x3-exact"; do python tools/shard_step_profile.py --world $R $P 2>&1 | tail -2; done
done | tee $O/shard_steps.txt
