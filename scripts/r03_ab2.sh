#!/bin/bash
# A/B of ab/lib$1.so and ab/lib$2.so: saturated stages + bench, interleaved, twice
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in "$@"; do
    echo "== $v"; DATOK_GPU_LIB=$PWD/ab/lib$v.so python scripts/big_stages.py 32 2>&1 | tail -1 | sed 's/.*per 16 MiB, us: //'
    DATOK_GPU_LIB=$PWD/ab/lib$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --parity-docs 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('   bench MB/s', j['value'], 'one', j['streams_1']['value'], 'walk1', j['streams_1']['stages_ms']['walk'], 'walk3', j['stages_ms']['walk'])
"
  done
done
