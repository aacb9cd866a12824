cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for a in "" "--no-rows-copy"; do echo -n "[$a] "; python tools/exact_step_profile.py --precisions proved --steps 150 $a 2>&1 | grep "ms per step" | cut -c36-80; done
done
for a in "" "--no-rows-copy"; do echo -n "k'=2561 [$a] "; python tools/exact_step_profile.py --precisions proved --steps 60 --k 2500 --k-prime 2561 $a 2>&1 | grep "ms per step" | cut -c36-80; done
cd /tmp && export TMPDIR=/tmp
for a in "" "--no-rows-copy"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o p -- python $GRAFT_REPO_ROOT/tools/exact_step_profile.py --precisions proved --steps 50 $a > /dev/null 2>&1
  f=$(find /tmp/prof_ab -name '*kernel_stats.csv' | head -1); echo "== [$a]"; python $GRAFT_REPO_ROOT/tools/kernel_stats_top.py "$f" 30 | grep -E "rows_kernel|direct_kernel|rescore_select" | cut -c1-160; rm -rf /tmp/prof_ab
done
