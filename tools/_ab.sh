python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "c4 or 16x16x64 or shape_fuzz or degenerate or one_product" 2>&1 | tail -5
for pr in fp32 f16x3 f16x1; do bash tools/wsplit_phases.sh run $pr 2>&1 | tail -2; python tools/score_bench.py --workload synthetic-16x16x64 --items 400000 --variants 0 --rounds 3 --check-items 64 --precision $pr 2>&1 | tail -2; done
