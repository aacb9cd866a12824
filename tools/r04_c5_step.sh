cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/c5step; mkdir -p $O
timeout 600 python tools/fuzz_fused_scans.py > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
for B in 32 128; do
timeout 900 python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --batch $B --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --no-recall --steps 20 --warmup 3 > $O/b$B.json 2> $O/b$B.err
python -c "import json; d=json.load(open('$O/b$B.json')); print('B=$B', d['ms_per_step'], d['value'], d['roofline']['kernel_ms'])"
done
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --no-recall --steps 20 --warmup 3 > $O/prof.json 2> $O/prof.err
python tools/kernel_stats_top.py $(find $O/prof -name '*kernel_stats.csv' | head -1) 24 | tee $O/kernel_split.txt
