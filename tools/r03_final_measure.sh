# End-of-round-3 re-measurement after the radix top-k work: bash tools/r03_final_measure.sh -> gpurun_out/r03_final2/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r03_final2; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/gputest.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o r03 -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/prof.log 2>&1
python tools/algorithms_bench.py --workload amzn-books > $O/algorithms_books.json 2> $O/algorithms_books.err
cat $O/gputest.log; ls $O $O/prof/*
