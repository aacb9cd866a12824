#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
O=/root/repo/gpurun_out/r06y; mkdir -p $O; rm -f $O/out.txt
python -m pytest tests -x -q -m gpu -k "coarse or avg or two_pass or config5 or component or naive or comb or f10 or f9 or f8 or algorithms or prefilter or fused" 2>&1 | tail -4 > $O/test.txt
python tools/algorithms_bench.py --workload amzn-books > $O/algo.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
for alg in MoLAvgTopK4000 MoLAvgTopK1000 MoLNaiveTopK100 MoLNaiveTopK5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms $alg > /dev/null 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo "== $alg" >> $O/out.txt; python /root/repo/tools/kernel_stats_top.py "$f" 40 | grep "coarse_scan_kernel<2, 2\|sublist" | cut -c1-70,100-170 >> $O/out.txt; rm -rf $O/prof
done
