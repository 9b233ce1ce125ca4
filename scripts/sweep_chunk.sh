cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in ${CHUNKS:-0 64 128 256 512}; do
  python bench.py --chunk $c --steps 30 --warmup 3 --no-cpu-baseline --parity-docs 64 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('chunk',$c,'value',j['value'],'ms',j['ms_per_step'],'walk_ms',j['roofline']['kernel_ms'],'lanes',j['walk']['lanes'],'Glookups/s',j['roofline']['Glookups_per_s'],'stages',j['stages_ms'])
"
done
