#!/bin/bash
cd /root/repo
O=gpurun_out/r06suite; mkdir -p $O
( time python -m pytest tests -q -m gpu --durations=40 ) > $O/gpu_suite.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
