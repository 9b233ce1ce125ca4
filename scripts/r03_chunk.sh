#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in 128 120 112 104; do
 for w in 8; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --parity-docs 32 --chunk $c --warm $w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('chunk $c warm $w: MB/s', j['value'], 'one', j['streams_1']['value'], 'walk1', j['streams_1']['stages_ms']['walk'], 'walk3', j['stages_ms']['walk'], 'repairs', j['walk']['repair_rounds'], 'lookups', j['roofline']['lookups_per_launch'])
"
 done
done
