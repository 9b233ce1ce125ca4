cd $GRAFT_REPO_ROOT
for st in 2 3 4 5 6 8 3; do
  python bench.py --streams $st --steps 60 --warmup 6 --no-cpu-baseline --parity-docs 16 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stages_ms']; print('streams',$st,'value',j['value'],'ms',j['ms_per_step'],'sum',round(sum(s.values()),4))
"
done
