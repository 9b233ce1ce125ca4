#!/bin/bash
# round 3: results on the host -- new tests, then the bench line's end_to_end block
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu -k "pipeline or result_fields or cabi_transduce or python_token_writer or cpp_mirror or config2" > gpurun_out/r03_e2e_tests.log 2>&1; echo "tests rc $?"; tail -5 gpurun_out/r03_e2e_tests.log
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --parity-docs 32 2>gpurun_out/r03_e2e_bench.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('MB/s', j['value'], 'one', j['streams_1']['value']); print(json.dumps(j['end_to_end'], indent=1))
"
tail -3 gpurun_out/r03_e2e_bench.err
