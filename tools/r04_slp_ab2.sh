cd $GRAFT_REPO_ROOT
for pass in 1 2; do
  for tag in base slp1; do
    lib=rails_amd/_ab/librails_amd_$tag.so; [ "$tag" = base ] && lib=rails_amd/librails_amd.so
    for wl in ml-20m ml-1m; do
      echo -n "[$tag] $wl f16x1 "; RAILS_AMD_LIBRARY=$lib python tools/score_bench.py --variants 0 --workload $wl --batch 32 --precision f16x1 --rounds 7 --reps 20 2>&1 | grep variant
    done
    echo -n "[$tag] books B=8 f16x1 "; RAILS_AMD_LIBRARY=$lib python tools/score_bench.py --variants 0 --workload amzn-books --batch 8 --precision f16x1 --rounds 5 --reps 5 2>&1 | grep variant
  done
done
