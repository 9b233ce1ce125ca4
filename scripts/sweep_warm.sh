cd $GRAFT_REPO_ROOT
for cw in "128 48" "128 32" "128 24" "128 16" "128 8" "128 48"; do
  set -- $cw
  python bench.py --chunk $1 --warm $2 --steps 60 --warmup 6 --no-cpu-baseline --parity-docs 64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stages_ms']; print('chunk',$1,'warm',$2,'value',j['value'],'ms',j['ms_per_step'],'walk',s['walk'],'repair',j['walk']['repair_rounds'])
"
done
