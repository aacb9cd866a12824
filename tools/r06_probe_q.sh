#!/bin/bash
# timelines of the dense fp32 route at B = 1 on 100 000 / 200 000 / 695 762 items of the amzn-books configuration and on ML-20M (which kernels, how long)
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r06_probe_q; mkdir -p $O
cd /tmp
for N in 100000 200000 695762; do
  rocprofv3 --kernel-trace --output-format csv -d $O/tl_$N -o t -- python /root/repo/tools/exact_step_profile.py --precisions fp32 --steps 60 --batch 1 --items $N > /dev/null 2>&1
done
rocprofv3 --kernel-trace --output-format csv -d $O/tl_ml -o t -- python /root/repo/tools/exact_step_profile.py --precisions fp32 --steps 60 --batch 2 --workload ml-20m > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/tl_mlp -o t -- python /root/repo/tools/exact_step_profile.py --precisions proved --steps 60 --batch 2 --workload ml-20m > /dev/null 2>&1
cd /root/repo
{ for N in 100000 200000 695762; do echo "== fp32 B=1 N=$N"; python tools/r06_timeline.py $(find $O/tl_$N -name '*kernel_trace.csv' | head -1) mol_score 2; done
  echo "== ml-20m fp32 B=2"; python tools/r06_timeline.py $(find $O/tl_ml -name '*kernel_trace.csv' | head -1) mol_score 2
  echo "== ml-20m proved B=2"; python tools/r06_timeline.py $(find $O/tl_mlp -name '*kernel_trace.csv' | head -1) F16Unit 2; } > $O/timelines.txt 2>&1
rm -rf $O/tl_*
