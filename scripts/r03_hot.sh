#!/bin/bash
# round 3: the LDS cache of hot cells -- GPU tests, then A/B against DATOK_NO_HOT=1 (saturated, one batch, three in flight)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
true
for r in 1 2; do
  for v in 1 0; do
    echo "== NO_HOT=$v saturated"; DATOK_NO_HOT=$v python scripts/big_stages.py 32 2>&1 | tail -2
  done
done
for r in 1 2; do
  for v in 1 0; do
    echo "== NO_HOT=$v bench"
    DATOK_NO_HOT=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --parity-docs 32 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('MB/s', j['value'], 'ms', j['ms_per_step'], 'one', j['streams_1']['value'], j['streams_1']['stages_ms'], '| 3:', j['stages_ms'])
"
  done
done
