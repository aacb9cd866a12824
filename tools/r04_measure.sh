# Round-4 measurement batch (one gpurun call): bash tools/r04_measure.sh -> gpurun_out/r04_final/   (copy what is judged into profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_final; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
# the headline step alone under rocprofv3 (every launch of the scoring kernel at B = 32), then the whole default command without the CPU leg
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o r04 -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --no-hr-parity > $O/bench_headline_under_profiler.json 2> $O/prof_headline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r04 -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1
# HBM traffic of the headline kernel (separate --pmc passes) and the SQ sets
bash tools/pmc_traffic.sh $O/pmc_traffic --workload amzn-books --batch 32 > $O/pmc_traffic_summary.txt 2>&1
PMC_EXTRA="--workload amzn-books --batch 32" bash tools/pmc.sh $O/pmc 0 > $O/pmc_fp32_summary.txt 2>&1
# config 5: one full 8-way shard, two-pass, with the recall phase; plain and under the profiler
python bench.py --workload synthetic-8x8x32 --two-pass 1000 --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/two_pass_125m.json 2> $O/two_pass_125m.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o r04c5 -- python bench.py --workload synthetic-8x8x32 --two-pass 1000 --no-recall --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/two_pass_125m_prof.json 2>> $O/two_pass_125m.err
# the coarse pass of config 5 on its own (no MoL index): ms per call, per-kernel split, fused == materialised check
bash tools/r04_c5_ab.sh final > $O/c5_coarse_pass.txt 2>&1; cp gpurun_out/c5ab_final/kernel_split.txt $O/c5_coarse_pass_kernel_split.txt 2>/dev/null
python bench.py --workload synthetic-8x8x32 --two-pass 1000 --batch 128 --no-recall --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/two_pass_125m_b128.json 2>> $O/two_pass_125m.err
# config 4: one full 8-way shard, exact
python bench.py --workload synthetic-16x16x64 --no-cpu-baseline --no-matrix --no-hr-parity --steps 5 --warmup 1 > $O/bench_c4_shard.json 2> $O/bench_c4.err
python bench.py --batch 128 --no-cpu-baseline --no-matrix --no-other-workloads --steps 10 --warmup 2 > $O/bench_b128.json 2> $O/bench_b128.err
# small-unit shell vs the 32x32x2 shells; small-corpus step timelines; per-rank step of the sharded path
bash tools/r04_small_sweep.sh > $O/small_sweep.txt 2>&1
for wl in ml-1m ml-20m; do rocprofv3 --kernel-trace -d $O/sc_$wl -o t --output-format csv -- python tools/small_corpus_trace.py --workload $wl > $O/sc_$wl.log 2>&1; python tools/step_timeline.py $(ls $O/sc_$wl/*kernel_trace.csv | head -1) 4 > $O/step_timeline_$wl.txt 2>&1; python tools/small_corpus_trace.py --workload $wl 2>&1 | grep "us/step" >> $O/step_timeline_$wl.txt; done
rm -f $O/shard_steps.txt; for R in 2 4 8; do python tools/shard_step_profile.py --world $R 2>&1 | tail -1 >> $O/shard_steps.txt; done
ls $O
