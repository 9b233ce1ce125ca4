#!/bin/bash
cd $GRAFT_REPO_ROOT
for w in 16 12 8 6 4; do
  python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --parity-docs 32 --warm $w 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('warm $w: MB/s', j['value'], 'one', j['streams_1']['value'], 'walk1', j['streams_1']['stages_ms']['walk'], 'walk3', j['stages_ms']['walk'], 'repairs', j['walk']['repair_rounds'], 'lookups', j['roofline']['lookups_per_launch'])
"
done
