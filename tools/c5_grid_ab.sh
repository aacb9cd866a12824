# config-5 two-pass step against the scans' grid cap: bash tools/c5_grid_ab.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/c5grid_$1; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -k "fused or avg or two_pass or large_batches or naive or comb or coarse" > $O/tests.txt 2>&1; tail -2 $O/tests.txt
timeout 600 python tools/fuzz_fused_scans.py > $O/fuzz.txt 2>&1; tail -1 $O/fuzz.txt
for G in 2048 768 1024 1536 4096; do
  export RAILS_SCAN_GRID=$G
  for B in 32 128; do
    timeout 900 python bench.py --workload synthetic-8x8x32 --two-pass 1000 --device-table --batch $B --no-cpu-baseline --no-matrix --no-other-workloads --no-fast-path --steps 10 --warmup 2 > $O/g${G}_b$B.json 2> $O/g${G}_b$B.err
    echo "grid $G B=$B $(python -c "import json,sys; print(json.load(open('$O/g${G}_b$B.json'))['ms_per_step'])")"
  done
done
