cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_final; mkdir -p $O
: > $O/shard_steps.txt
for R in 2 4 8; do for P in fp32 proved-global f16x3-exact; do python tools/shard_step_profile.py --world $R --precision $P 2>&1 | grep -v amdgpu.ids >> $O/shard_steps.txt; done; done
grep "ms/step" $O/shard_steps.txt | cut -c1-120
bash tools/r05_measure.sh final 2>&1 | tail -45
