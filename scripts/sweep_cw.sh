cd $GRAFT_REPO_ROOT
for cw in "128 64" "128 48" "128 40" "96 48" "96 40" "64 40" "192 48" "256 48"; do
  set -- $cw
  python bench.py --chunk $1 --warm $2 --steps 30 --warmup 3 --no-cpu-baseline --parity-docs 64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stages_ms']; print('chunk',$1,'warm',$2,'value',j['value'],'ms',j['ms_per_step'],'start',s['spec_start'],'walk',s['walk'],'repair',j['walk']['repair_rounds'])
"
done
