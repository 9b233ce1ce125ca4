#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_exact_and_replay.py -x -q -m gpu > gpurun_out/r02/t_exact.log 2>&1; echo "exact rc=$?"
tail -4 gpurun_out/r02/t_exact.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu --deselect tests/test_exact_and_replay.py > gpurun_out/r02/t_all.log 2>&1; echo "all rc=$?"
tail -6 gpurun_out/r02/t_all.log
timeout -k 10 200 python scripts/soak.py 20 1000 > gpurun_out/r02/soak.log 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/r02/soak.log
timeout -k 10 300 python scripts/sweep2.py 64,96,128,192,256 1,2,3,4 > gpurun_out/r02/sweep.log 2>&1; echo "sweep rc=$?"; cat gpurun_out/r02/sweep.log
