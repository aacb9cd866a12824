# the approximate algorithms of eval_batch.py on the default build and on each variant under rails_amd/_ab, same box: bash tools/algo_ab.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/algo_$1; mkdir -p $O
for rep in 1 2; do for lib in default $(ls rails_amd/_ab/ 2>/dev/null); do
  [ $lib = default ] && unset RAILS_AMD_LIBRARY || export RAILS_AMD_LIBRARY=$PWD/rails_amd/_ab/$lib
  python tools/algorithms_bench.py --workload amzn-books --algorithms MoLNaiveTopK5,MoLNaiveTopK100,MoLAvgTopK200,MoLAvgTopK4000,MoLCombTopK50_500 > $O/${lib}_$rep.json 2> $O/${lib}_$rep.err
  python -c "
import json; r=json.load(open('$O/${lib}_$rep.json'))['rows']; print('$lib', ' '.join(f\"{x['algorithm'][3:]}={x['BatchTimeMsMedian']:.3f}\" for x in r))"
done; done
