#!/bin/bash
# the component select scan at k_g = 100: the default block against the scalar-mask block (RAILS_COMP_SELECT=1), and without candidates (RAILS_COMP_DEBUG=1)
cd "$(dirname "$0")/.." || exit 1
O=/root/repo/gpurun_out/r06t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in default sel nohits; do
  case $v in default) E="";; sel) E="RAILS_COMP_SELECT=1";; nohits) E="RAILS_COMP_DEBUG=1";; esac
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms MoLNaiveTopK100 > /dev/null 2>&1
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); echo "== $v" >> $O/out.txt; python /root/repo/tools/kernel_stats_top.py "$f" 6 | cut -c1-70,100-170 >> $O/out.txt; rm -rf $O/prof_$v
done
