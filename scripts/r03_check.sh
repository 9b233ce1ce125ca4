#!/bin/bash
# full GPU tier + one bench line (round 3 working check)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/r03_tests.log 2>&1; echo "gpu tests rc $?"; tail -4 gpurun_out/r03_tests.log
python bench.py --steps 40 --warmup 5 --cpu-seconds 3 --parity-docs 32 2>gpurun_out/r03_bench.err > gpurun_out/r03_bench.json; echo "bench rc $?"
python - <<'PY'
import json
for l in open('gpurun_out/r03_bench.json'):
    if l.startswith('{'):
        j = json.loads(l)
        print('MB/s', j['value'], 'ms', j['ms_per_step'], 'one', j['streams_1']['value'], 'frac', j['roofline']['frac'])
        print('stages3', j['stages_ms']); print('stages1', j['streams_1']['stages_ms'])
        print({k: v for k, v in j['end_to_end'].items() if not k.endswith('what')})
PY
tail -2 gpurun_out/r03_bench.err
