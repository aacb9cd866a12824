#!/bin/bash
# rocprofv3 --kernel-trace --stats of four algorithms of the eval_batch.py table at HEAD -> top kernels of each
cd "$(dirname "$0")/.." || exit 1
O=/root/repo/gpurun_out/r06algo; mkdir -p $O; rm -f $O/out.txt
cd /tmp && export TMPDIR=/tmp
for alg in MoLAvgTopK200 MoLAvgTopK4000 MoLNaiveTopK5 MoLNaiveTopK100 MoLCombTopK100_1000; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms $alg > /dev/null 2>&1
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo "== $alg (23 timed + warm-up calls; kernel, calls, average, share of the run incl. index builds)" >> $O/out.txt
  python /root/repo/tools/kernel_stats_top.py "$f" 30 | grep -v "index_build\|component_build\|index_rows\|coarse_build\|pack_gate\|copyBuffer\|at::native\|prefilter_build\|elementwise\|fillBuffer" | cut -c1-72,100-170 >> $O/out.txt; rm -rf $O/prof
done
