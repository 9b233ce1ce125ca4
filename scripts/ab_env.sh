#!/bin/bash
# A/B one build with and without an environment switch, interleaved on one box.
# usage: gpurun -- bash scripts/ab_env.sh VAR=VALUE [bench args]
cd $GRAFT_REPO_ROOT
kv=$1; shift
for r in 1 2 3; do
  for v in A B; do
    if [ $v = A ]; then pre="env $kv"; else pre="env"; fi
    $pre python bench.py --steps 40 --warmup 5 --no-cpu-baseline --parity-docs 32 "$@" 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('$v', 'MB/s', j['value'], 'ms', j['ms_per_step'], 'lookups', j['roofline']['lookups_per_launch'], j['stages_ms'])
"
  done
done
