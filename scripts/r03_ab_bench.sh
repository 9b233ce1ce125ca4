#!/bin/bash
# bench A/B of libraries in ab/, interleaved, three rounds
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in "$@"; do
    DATOK_GPU_LIB=$PWD/ab/lib$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --parity-docs 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v MB/s', j['value'], 'ms', j['ms_per_step'], 'one', j['streams_1']['value'], j['streams_1']['ms_per_step'])
"
  done
done
