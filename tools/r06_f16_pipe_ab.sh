#!/bin/bash
# round 6, experiment (b) on the f16x3 first pass (TIGHT stream, 8x8x32): same-box alternating A/B of the variant builds under rocprofv3
# (kernel average of mol_score_staged_kernel<f16x3::F16Unit...> over 100 steady-state steps each, two rounds)
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/r06pipe; mkdir -p $O
for rep in 1 2; do
for tag in base ypipe xpipe xypipe pf4; do
  lib=/root/repo/rails_amd/_ab/librails_amd_r06_$tag.so; [ $tag = base ] && lib=/root/repo/rails_amd/librails_amd.so
  RAILS_AMD_LIBRARY=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_$rep -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 100 > $O/${tag}_$rep.log 2>&1
  f=$(find $O/prof_${tag}_$rep -name "*kernel_stats.csv" | head -1)
  echo "$tag rep$rep $(grep 'F16Unit' $f | head -1 | awk -F, '{print $(NF-5), $(NF-4)}') $(grep -v amdgpu $O/${tag}_$rep.log | grep proved | cut -c1-75)" >> $O/summary.txt
  rm -rf $O/prof_${tag}_$rep
done
done
