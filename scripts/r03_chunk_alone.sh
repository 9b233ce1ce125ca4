#!/bin/bash
# chunk size against one batch alone / three in flight (bench corpus)
cd $GRAFT_REPO_ROOT
for c in 48 64 80 96 128; do
python bench.py --chunk $c --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --parity-docs 16 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('chunk $c: three in flight', j['value'], 'alone', j['streams_1']['value'], 'walk alone', j['streams_1']['stages_ms']['walk'], 'sym', j['streams_1']['stages_ms']['symbolize'], 'rounds', j['walk']['repair_rounds'])
"
done
