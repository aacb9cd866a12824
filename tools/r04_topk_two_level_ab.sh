cd $GRAFT_REPO_ROOT
for pass in 1 2; do for tag in base tl512; do
  lib=rails_amd/_ab/librails_amd_$tag.so; [ "$tag" = base ] && lib=rails_amd/librails_amd.so
  for nk in "695762 1000" "695762 1600" "200000 1000" "200000 2561" "200000 4096" "1200000 600" "100000 2000"; do set -- $nk
    echo -n "[$tag] "; RAILS_AMD_LIBRARY=$lib python tools/topk_bench.py --rows 32 --n $1 --k $2 --dist narrow 2>&1 | tail -1
  done; done; done
