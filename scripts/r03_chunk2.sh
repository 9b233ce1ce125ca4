#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2; do
for c in 128 104; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --parity-docs 32 --chunk $c 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('chunk $c: MB/s', j['value'], 'one', j['streams_1']['value']); print('   3:', j['stages_ms']); print('   1:', j['streams_1']['stages_ms'])
"
done
done
