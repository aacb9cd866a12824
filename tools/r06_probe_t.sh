#!/bin/bash
# the select scans with many candidates: default, without candidates (RAILS_COMP_DEBUG=1), with candidates but without the appends' global atomics (=2: wrong results, time only)
cd "$(dirname "$0")/.." || exit 1
O=/root/repo/gpurun_out/r06t; mkdir -p $O; rm -f $O/out.txt
cd /tmp && export TMPDIR=/tmp
for alg in MoLAvgTopK4000 MoLAvgTopK1000; do
for v in default nohits noatomics; do
  case $v in default) E="A=1";; nohits) E="RAILS_COMP_DEBUG=1";; noatomics) E="RAILS_COMP_DEBUG=2";; esac
  env $E rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms $alg > /dev/null 2>&1
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); echo "== $alg $v" >> $O/out.txt; python /root/repo/tools/kernel_stats_top.py "$f" 40 | grep "coarse_scan_kernel<2, 2" | cut -c1-70,100-170 >> $O/out.txt; rm -rf $O/prof_$v
done; done
