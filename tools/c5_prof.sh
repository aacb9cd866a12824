# kernel split of the config-5 two-pass step under rocprofv3: bash tools/c5_prof.sh <tag> [library]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/c5prof_$1; mkdir -p $O
[ -n "$2" ] && export RAILS_AMD_LIBRARY=$PWD/rails_amd/_ab/$2
for B in 32 128; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$B -o c5 -- python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --batch $B --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/b$B.json 2> $O/b$B.err
  python tools/kernel_stats_top.py $(find $O/p$B -name "*kernel_stats.csv" | head -1) 8 | tee $O/top$B.txt
done
