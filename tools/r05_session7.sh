#!/bin/bash
# small corpora: where the proved flow pays, and what the dense fp32 step spends at 16 k items
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/s17; mkdir -p $O
: > $O/crossover.txt
for W in ml-1m ml-20m; do
  python tools/exact_step_profile.py --workload $W --precisions fp32,proved,f16x3 --min-items 0 --steps 200 --width 211 2>&1 | grep -v amdgpu.ids >> $O/crossover.txt
done
cat $O/crossover.txt
cd /tmp && export TMPDIR=/tmp
for P in fp32 proved; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$P -o p -- python $GRAFT_REPO_ROOT/tools/exact_step_profile.py --items 16384 --precisions $P --min-items 0 --steps 100 > $GRAFT_REPO_ROOT/$O/prof_$P.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof_$P -name '*kernel_stats.csv' | head -1)
  echo "== $P"; python $GRAFT_REPO_ROOT/tools/kernel_stats_top.py "$f" 2>&1 | head -14
  grep "ms per step" $GRAFT_REPO_ROOT/$O/prof_$P.log
done
for P in fp32 proved; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof20_$P -o p -- python $GRAFT_REPO_ROOT/tools/exact_step_profile.py --workload ml-20m --width 211 --precisions $P --min-items 0 --steps 100 > $GRAFT_REPO_ROOT/$O/prof20_$P.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof20_$P -name '*kernel_stats.csv' | head -1)
  echo "== ml-20m $P"; python $GRAFT_REPO_ROOT/tools/kernel_stats_top.py "$f" 2>&1 | head -14
  grep "ms per step" $GRAFT_REPO_ROOT/$O/prof20_$P.log
done
