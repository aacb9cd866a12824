#!/bin/bash
cd /root/repo
O=gpurun_out/r06j; mkdir -p $O
# PMC pass 1 (stall counters) of the f16x3 first pass: shipped stream and the stage-Y pipelined variant
for tag in base ypipe; do
  lib=/root/repo/rails_amd/_ab/librails_amd_r06_$tag.so; [ $tag = base ] && lib=/root/repo/rails_amd/librails_amd.so
  RAILS_AMD_LIBRARY=$lib PMC_EXTRA='--precision f16x3' bash tools/pmc.sh $O/pmc_$tag 0 > $O/pmc_$tag.txt 2>&1
done
python bench.py > $O/bench.json 2> $O/bench.err
