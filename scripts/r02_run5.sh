#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t_bits.log 2>&1; echo "tests rc=$?"
tail -4 gpurun_out/r02/t_bits.log
timeout -k 10 200 python scripts/sweep2.py 64,128,256 1,2,3,4 2>&1 | tail -4
DATOK_LDS_BITS=0 timeout -k 10 200 python scripts/sweep2.py 128 1,3 2>&1 | tail -2
timeout -k 10 200 python scripts/robust.py 2>&1 | tail -5
