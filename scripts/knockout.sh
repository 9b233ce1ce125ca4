# Marginal cost of each stage with several batches in flight: needs a build with -DDTK_EXPERIMENTS
#   make -C datok_amd/csrc CXXFLAGS='-O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DDTK_EXPERIMENTS'
# skip mask: 1 symbolize, 2 start+link, 4 walk+memset, 8 compact (from the second run of a batch on)
cd $GRAFT_REPO_ROOT
for sk in 0 1 2 4 8 6 7 15 0; do
  DATOK_EXP_SKIP=$sk python bench.py --streams ${STREAMS:-3} --steps 60 --warmup 6 --no-cpu-baseline --parity-docs 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stages_ms']; print('skip',$sk,'value',j['value'],'ms',j['ms_per_step'], s)
"
done
