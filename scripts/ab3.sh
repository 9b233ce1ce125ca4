#!/bin/bash
# A/B/C/D several experimental builds (ab/libX.so) on one box, interleaved. usage: ab3.sh "A B C" [bench args]
cd $GRAFT_REPO_ROOT
libs=$1; shift
for r in 1 2; do
  for v in $libs; do
    DATOK_GPU_LIB=$PWD/ab/lib$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --parity-docs 0 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'MB/s', j['value'], 'ms', j['ms_per_step'], j['stages_ms'])
"
  done
done
