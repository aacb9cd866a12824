#!/bin/bash
# config 4 (16x16x64), the default exact path's dominant kernel -- the UPPER first pass of the team kernel: rocprofv3 stats of the shard bench,
# PMC traffic and SQ counters at 400 k items
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r05_c4; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c4 -- python bench.py --workload synthetic-16x16x64 --no-cpu-baseline --no-matrix --no-other-workloads --no-hr-parity --no-fast-path --steps 5 --warmup 1 > $O/bench_c4_under_profiler.json 2> $O/prof.err
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_c4.csv; python tools/kernel_stats_top.py "$f" 14 > $O/kernel_stats_c4_top.txt; rm -rf $O/prof
bash tools/pmc_traffic.sh $O/pmc_traffic --workload synthetic-16x16x64 --items 400000 --batch 32 --precision f16x3 --upper > $O/pmc_traffic_c4_upper_summary.txt 2>&1
PMC_EXTRA="--workload synthetic-16x16x64 --items 400000 --batch 32 --precision f16x3 --upper" bash tools/pmc.sh $O/pmc 0 > $O/pmc_c4_upper_summary.txt 2>&1
rm -rf $O/pmc_traffic $O/pmc
cat $O/kernel_stats_c4_top.txt | cut -c1-170; cat $O/pmc_traffic_c4_upper_summary.txt | tail -4; tail -22 $O/pmc_c4_upper_summary.txt
python -c "
import json; d=json.load(open('$O/bench_c4_under_profiler.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['proved']['proved_calls'], d['proved']['timed_calls'])"
