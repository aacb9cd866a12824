# config-5 two-pass step (125 M-item shard, K' = 1000) at B = 32 / 128 for each library variant under rails_amd/_ab (plus the
# default build), and the coarse-scan parity subset on the default build: bash tools/c5_scan_ab.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/c5_$1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "fused or avg or two_pass or large_batches or naive or comb or coarse" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 600 python tools/fuzz_fused_scans.py > $O/fuzz.txt 2>&1; tail -2 $O/fuzz.txt
for lib in default $(ls rails_amd/_ab/ 2>/dev/null); do
  [ $lib = default ] && unset RAILS_AMD_LIBRARY || export RAILS_AMD_LIBRARY=$PWD/rails_amd/_ab/$lib
  for B in 32 128; do
    timeout 900 python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --batch $B --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/${lib}_b$B.json 2> $O/${lib}_b$B.err
    echo "$lib B=$B $(python -c "import json,sys; print(json.load(open('$O/${lib}_b$B.json'))['ms_per_step'])")"
  done
done
