#!/bin/bash
# A/B two builds of libdatok_gpu.so on ONE box, interleaved (boxes differ by several percent).
# usage (here): scripts/ab_prepare.sh   then   gpurun -- bash scripts/ab.sh [bench args]
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in A B; do
    DATOK_GPU_LIB=$PWD/ab/lib$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --parity-docs 32 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'MB/s', j['value'], 'ms', j['ms_per_step'], j['stages_ms'])
"
  done
done
