cd /root/repo; O=gpurun_out/r06m; mkdir -p $O
for sc in 3 4 6 8; do python tools/r06_scale_probe.py $sc 40 1000 2>&1 | grep scale >> $O/scale.txt; done
cat $O/scale.txt
