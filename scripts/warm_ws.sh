cd $GRAFT_REPO_ROOT
for cfg in "0 0" "1 0" "1 8" "1 16" "1 24" "2 8"; do
  set -- $cfg
  DATOK_WARM_WS=$1 DATOK_WARM_MIN=$2 python bench.py --streams ${STREAMS:-3} --steps 60 --warmup 6 --no-cpu-baseline --parity-docs 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stages_ms']; print('ws $1 min $2','value',j['value'],'ms',j['ms_per_step'],'start',s['spec_start'],'walk',s['walk'],'lookups',j['roofline']['lookups_per_launch'],'repair',j['walk']['repair_rounds'])
"
  DATOK_WARM_WS=$1 DATOK_WARM_MIN=$2 python scripts/configs.py 2>&1 | grep config3 | sed 's/.*GPU/   config3 GPU/'
done
