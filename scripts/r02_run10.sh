#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t_dev.log 2>&1; echo "tests rc=$?"
tail -6 gpurun_out/r02/t_dev.log
DATOK_DEV_ROUNDS=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or edge or rich or longer or long" > gpurun_out/r02/t_dev2.log 2>&1; echo "tests(dev rounds forced) rc=$?"
tail -3 gpurun_out/r02/t_dev2.log
timeout -k 10 200 python scripts/robust.py 2>&1 | tail -4
DATOK_DEV_ROUNDS=0 timeout -k 10 200 python scripts/robust.py 2>&1 | tail -4
