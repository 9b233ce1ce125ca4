#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_exact_and_replay.py -x -q -m gpu > gpurun_out/r02/t_fast.log 2>&1; echo "tests rc=$?"
tail -12 gpurun_out/r02/t_fast.log
timeout -k 10 200 python scripts/sweep2.py 128 1,2,3,4 2>&1 | tail -1
timeout -k 10 200 python scripts/robust.py 2>&1 | tail -4
