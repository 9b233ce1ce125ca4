cd $GRAFT_REPO_ROOT
for st in 1 2 3; do
  python bench.py --streams $st --steps 60 --warmup 6 --no-cpu-baseline --parity-docs 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('streams',$st,'MB/s',j['value'],'ms',j['ms_per_step'],'walk_ms',j['roofline']['kernel_ms'],'frac',j['roofline']['frac'],j['stages_ms'])
"
done
