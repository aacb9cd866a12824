cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_final2; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r02 -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-matrix > $O/prof.log 2>&1
PMC_EXTRA="--precision f16x3" bash tools/pmc.sh $O/pmc_f16 0 > $O/pmc_f16.txt 2>&1
bash tools/pmc.sh $O/pmc_fp32 0 > $O/pmc_fp32.txt 2>&1
PMC_EXTRA="--precision f16x3 --workload synthetic-16x16x64 --items 400000" bash tools/pmc.sh $O/pmc_c4_f16 0 > $O/pmc_c4_f16.txt 2>&1
python bench.py --workload synthetic-16x16x64 --no-cpu-baseline --no-matrix --steps 5 --warmup 1 > $O/bench_c4_shard.json 2> $O/bench_c4.err
for R in 2 4 8; do python tools/shard_step_profile.py --world $R >> $O/shard_steps.txt 2>&1; python tools/shard_step_profile.py --world $R --precision f16x3 >> $O/shard_steps.txt 2>&1; done
python tools/algorithms_bench.py --workload amzn-books > $O/algorithms_books.json 2> $O/algorithms_books.err
python tools/hstu_bench.py > $O/hstu.json 2> $O/hstu.err
ls $O
