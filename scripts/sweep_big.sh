# Stage times of one large batch (256 MiB) for several chunk sizes, one stream.  LIBS="A B": ab/libX.so builds
cd $GRAFT_REPO_ROOT
for v in ${LIBS:-cur}; do
for cw in "128 48" "256 48" "512 48" "1024 48"; do
  set -- $cw
  lib=$PWD/ab/lib$v.so; [ "$v" = cur ] && lib=$PWD/datok_amd/libdatok_gpu.so
  DATOK_GPU_LIB=$lib python bench.py --docs ${DOCS:-65536} --chunk $1 --warm $2 --streams ${STREAMS:-1} --steps 12 --warmup 3 --no-cpu-baseline --parity-docs 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); s=j['stages_ms']; print('$v chunk',$1,'warm',$2,'value',j['value'],'ms',j['ms_per_step'],s)
"
done
done
