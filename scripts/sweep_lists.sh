# event lists on/off for several batch sizes (one stream and three)
cd $GRAFT_REPO_ROOT
for docs in 4096 8192 16384 32768; do
 for st in 1 3; do
  for el in 0 1; do
  DATOK_EV_LISTS=$el python bench.py --docs $docs --streams $st --steps 20 --warmup 4 --no-cpu-baseline --parity-docs 16 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stages_ms']; print('docs',$docs,'streams',$st,'lists',$el,'value',j['value'],'ms',j['ms_per_step'],'walk',s['walk'],'chunk',j['walk']['chunk_bytes'])
"
  done
 done
done
