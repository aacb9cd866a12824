cd $GRAFT_REPO_ROOT
for pass in 1 2; do for tag in base sortbig; do
  lib=rails_amd/_ab/librails_amd_$tag.so; [ "$tag" = base ] && lib=rails_amd/librails_amd.so
  for nk in "3200 3200" "3200 1000" "4096 2561" "6400 6400" "1400 1400" "16000 2000" "8000 600"; do set -- $nk
    echo -n "[$tag] "; RAILS_AMD_LIBRARY=$lib python tools/topk_bench.py --rows 32 --n $1 --k $2 --dist narrow 2>&1 | tail -1
  done; done; done
