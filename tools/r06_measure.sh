#!/bin/bash
# Round-6 measurement batch (one gpurun call): bash tools/r06_measure.sh [tag] -> gpurun_out/r06_<tag>/   (copy what is judged into profiles/)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=gpurun_out/r06_${1:-final}; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench_wall.txt; echo "bench rc=$?"; tail -3 $O/bench_wall.txt
# the headline step alone under rocprofv3 (every launch of the proved step at B = 32)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o r06 -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --no-hr-parity --no-weights-sweep > $O/bench_headline_under_profiler.json 2> $O/prof_headline.err
f=$(find $O/prof_headline -name '*kernel_stats.csv' | head -1); cp "$f" $O/kernel_stats_headline.csv; python tools/kernel_stats_top.py "$f" 24 > $O/kernel_stats_headline_top.txt; rm -rf $O/prof_headline
# timelines of the proved step (launch order, kernel time incl. its dispatch gap): C3, B = 8, C2 (ML-20M), an R = 8 shard
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/tl_c3 -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 60 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/tl_b8 -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 60 --batch 8 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/tl_c2 -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 100 --workload ml-20m > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /root/repo/$O/tl_r8 -o t -- python /root/repo/tools/shard_step_profile.py --world 8 --precision proved-global --steps 100 > /dev/null 2>&1
cd /root/repo
{ echo "== amzn-books, B = 32 (C3)"; python tools/r06_timeline.py $(find $O/tl_c3 -name '*kernel_trace.csv' | head -1);
  echo "== amzn-books, B = 8"; python tools/r06_timeline.py $(find $O/tl_b8 -name '*kernel_trace.csv' | head -1);
  echo "== ML-20M, B = 32 (C2)"; python tools/r06_timeline.py $(find $O/tl_c2 -name '*kernel_trace.csv' | head -1) F16Unit 2;
  echo "== one of 8 shards of amzn-books (86 971 items), global proof, the all-gather replaced by a device copy"; python tools/r06_timeline.py $(find $O/tl_r8 -name '*kernel_trace.csv' | head -1); } > $O/step_timelines.txt 2>&1
rm -rf $O/tl_c3 $O/tl_b8 $O/tl_c2 $O/tl_r8
# step times outside the profiler
{ python tools/exact_step_profile.py --precisions proved,fp32 --steps 200; python tools/exact_step_profile.py --precisions proved --steps 200 --batch 8;
  python tools/exact_step_profile.py --precisions proved,fp32 --steps 400 --workload ml-20m; python tools/exact_step_profile.py --precisions proved,fp32 --steps 400 --workload ml-1m;
  for R in 8 4 2; do python tools/shard_step_profile.py --world $R --precision proved-global; python tools/shard_step_profile.py --world $R --precision proved-global --pipeline; python tools/shard_step_profile.py --world $R; done; } 2>&1 | grep -v amdgpu | cut -c1-220 > $O/step_times.txt
python tools/algorithms_bench.py --workload amzn-books > $O/algorithms_amzn_books.json 2> /dev/null
# the sharded path of bench.py in a process group of ONE rank over backend nccl (RCCL's all-gather / all-reduce / barrier on device tensors, exchange stream, merge + global verdict)
RAILS_BENCH_TEST_ONE_RANK_EXCHANGE=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-matrix --no-hr-parity --no-weights-sweep > $O/bench_one_rank_rccl.json 2> $O/bench_one_rank_rccl.err; echo "one-rank rccl bench rc=$?"
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; tail -5 $O/gpu_tests.txt
