#!/bin/bash
# round 6: the proved path with the fused tail against the dense fp32 kernels over many batches (weights x tables x query batches)
cd /root/repo; O=gpurun_out/r06sweep; mkdir -p $O
python tools/r05_proved_sweep.py --batches 100 --out $O/proved_sweep_books.json > $O/books.txt 2>&1
python tools/r05_proved_sweep.py --workload ml-20m --batches 20 --out $O/proved_sweep_ml20m.json > $O/ml20m.txt 2>&1
python tools/r05_proved_sweep.py --items 150000 --batches 20 --out $O/proved_sweep_150k.json > $O/s150k.txt 2>&1
python tools/r05_proved_sweep.py --workload synthetic-16x16x64 --weights 2 --tables 1 --batches 20 --out $O/proved_sweep_c4.json > $O/c4.txt 2>&1
for f in books ml20m s150k c4; do tail -n 1 $O/$f.txt; done
