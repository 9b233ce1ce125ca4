#!/bin/bash
# round 3: more waves per SIMD for the first-pass walk (72 / 64 VGPRs, 16-code windows)
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for v in O0 O7w32 O7w16 O8w16; do
    echo "== $v"; DATOK_GPU_LIB=$PWD/ab/lib$v.so python scripts/big_stages.py 32 2>&1 | tail -1 | sed 's/.*per 16 MiB, us: //'
    DATOK_GPU_LIB=$PWD/ab/lib$v.so python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --parity-docs 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('   bench MB/s', j['value'], 'one', j['streams_1']['value'], 'walk1', j['streams_1']['stages_ms']['walk'], 'walk3', j['stages_ms']['walk'])
"
  done
done
