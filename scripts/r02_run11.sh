#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipeline or render or cli" > gpurun_out/r02/t_pipe.log 2>&1; echo "tests rc=$?"
tail -8 gpurun_out/r02/t_pipe.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r02/bench2.json 2> gpurun_out/r02/bench2.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r02/bench2.json').read().strip().split('\n')[-1])
print(j['value'], j['ms_per_step'], j['streams_1'], j['end_to_end'])
PY
tail -3 gpurun_out/r02/bench2.err
