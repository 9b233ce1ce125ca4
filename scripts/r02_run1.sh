#!/bin/bash
# round 2, first GPU call: the new tests first (fail fast), then the whole GPU suite, the bench line and a kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_exact_and_replay.py -x -q -m gpu > gpurun_out/r02/t_exact.log 2>&1; echo "exact rc=$?"
tail -5 gpurun_out/r02/t_exact.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=12 > gpurun_out/r02/t_parity.log 2>&1; echo "parity rc=$?"
tail -20 gpurun_out/r02/t_parity.log
timeout -k 10 300 python bench.py > gpurun_out/r02/bench1.json 2> gpurun_out/r02/bench1.err; echo "bench rc=$?"
cat gpurun_out/r02/bench1.json; tail -3 gpurun_out/r02/bench1.err
