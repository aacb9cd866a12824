#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=/root/repo/gpurun_out/r06x; mkdir -p $O; rm -f $O/out.txt
cd /tmp && export TMPDIR=/tmp
for alg in MoLAvgTopK200 MoLAvgTopK4000 MoLNaiveTopK5 MoLCombTopK5_200; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$alg -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms $alg > /dev/null 2>&1
  f=$(find $O/prof_$alg -name "*kernel_stats.csv" | head -1); echo "== $alg" >> $O/out.txt; python /root/repo/tools/kernel_stats_top.py "$f" 16 | cut -c1-70,100-170 >> $O/out.txt; rm -rf $O/prof_$alg
done
