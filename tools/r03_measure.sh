# Round-3 final measurement batch (one gpurun call): bash tools/r03_measure.sh -> gpurun_out/r03_final/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_final; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03 -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-matrix > $O/prof.log 2>&1
python bench.py --workload synthetic-16x16x64 --no-cpu-baseline --no-matrix --steps 5 --warmup 1 > $O/bench_c4_shard.json 2> $O/bench_c4.err
rm -f $O/shard_steps.txt
for R in 2 4 8; do for pr in fp32 f16x3 f16-exact; do p=""; [ $pr != fp32 ] && p="--precision $pr"; python tools/shard_step_profile.py --world $R $p 2>&1 | tail -1 >> $O/shard_steps.txt; python tools/shard_step_profile.py --world $R $p --pipeline 2>&1 | tail -1 >> $O/shard_steps.txt; done; done
python tools/algorithms_bench.py --workload amzn-books > $O/algorithms_books.json 2> $O/algorithms_books.err
python tools/hstu_bench.py > $O/hstu_encoder.json 2> $O/hstu.err
python bench.py --batch 128 --no-cpu-baseline --no-matrix --no-other-workloads --steps 10 --warmup 2 > $O/bench_b128.json 2> $O/bench_b128.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o r03c5 -- python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/two_pass_125m.json 2> $O/two_pass_125m.err
python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --batch 32 --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/two_pass_125m_noprof.json 2>> $O/two_pass_125m.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5_b128 -o r03c5b128 -- python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --batch 128 --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/two_pass_125m_b128_prof.json 2> $O/two_pass_125m_b128.err
python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --batch 128 --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/two_pass_125m_b128.json 2>> $O/two_pass_125m_b128.err
ls $O
