cd $GRAFT_REPO_ROOT
for cw in "128 48" "96 48" "160 48" "192 48" "256 48" "128 40" "128 32" "128 48"; do
  set -- $cw
  python bench.py --chunk $1 --warm $2 --streams ${STREAMS:-3} --steps 60 --warmup 6 --no-cpu-baseline --parity-docs 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stages_ms']; print('chunk',$1,'warm',$2,'value',j['value'],'ms',j['ms_per_step'],'start',s['spec_start'],'walk',s['walk'],'repair',j['walk']['repair_rounds'])
"
done
