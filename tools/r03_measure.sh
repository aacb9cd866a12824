cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_final; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03 -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-matrix > $O/prof.log 2>&1
bash tools/pmc.sh $O/pmc_fp32 0 > $O/pmc_fp32.txt 2>&1
PMC_EXTRA="--precision f16x3" bash tools/pmc.sh $O/pmc_f16 0 > $O/pmc_f16.txt 2>&1
for pr in fp32 f16x3 f16x1; do PMC_EXTRA="--precision $pr --workload synthetic-16x16x64 --items 400000" bash tools/pmc.sh $O/pmc_c4_$pr 0 > $O/pmc_c4_$pr.txt 2>&1; done
python bench.py --workload synthetic-16x16x64 --no-cpu-baseline --no-matrix --steps 5 --warmup 1 > $O/bench_c4_shard.json 2> $O/bench_c4.err
for R in 2 4 8; do for pr in fp32 f16x3 f16-exact; do p=""; [ $pr != fp32 ] && p="--precision $pr"; python tools/shard_step_profile.py --world $R $p 2>&1 | tail -1 >> $O/shard_steps.txt; python tools/shard_step_profile.py --world $R $p --pipeline 2>&1 | tail -1 >> $O/shard_steps.txt; done; done
for pr in fp32 f16x3 f16x1; do bash tools/wsplit_phases.sh run $pr 2>&1 | tail -2 >> $O/wsplit_phases.txt; done
python tools/algorithms_bench.py --workload amzn-books > $O/algorithms_books.json 2> $O/algorithms_books.err
ls $O
