#!/bin/bash
# Naive / Comb with the seen-id filter inside the candidates' selection launch (top k + width instead of the full ranking): tests + algorithms bench + kernel stats
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/r06r; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "candidates or naive or comb or f10 or component or filter or algorithms or sort_rows" 2>&1 | tail -5 > $O/test.txt
python tools/algorithms_bench.py --workload amzn-books --algorithms MoLNaiveTopK5,MoLNaiveTopK10,MoLNaiveTopK50,MoLNaiveTopK100,MoLCombTopK5_200,MoLCombTopK50_500,MoLCombTopK100_1000 > $O/algo.json 2> $O/algo.err
cd /tmp && export TMPDIR=/tmp
for alg in MoLNaiveTopK100 MoLCombTopK100_1000; do
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$O/prof_$alg -o t -- python /root/repo/tools/algorithms_bench.py --workload amzn-books --algorithms $alg > /dev/null 2>&1
f=$(find /root/repo/$O/prof_$alg -name "*kernel_stats.csv" | head -1); python /root/repo/tools/kernel_stats_top.py "$f" 18 > /root/repo/$O/top_$alg.txt; rm -rf /root/repo/$O/prof_$alg
done
