#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t_bits.log 2>&1; echo "tests rc=$?"
tail -25 gpurun_out/r02/t_bits.log
timeout -k 10 200 python scripts/sweep2.py 64,128,256 1,3 2>&1 | tail -4
