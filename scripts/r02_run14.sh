#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t_sym.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r02/t_sym.log
timeout -k 10 100 python scripts/sweep2.py 128 1,3 2>&1 | tail -1
timeout -k 10 120 python scripts/soak.py 12 5000 2>&1 | tail -1
timeout -k 10 200 python scripts/tiny_docs.py 2>&1 | tail -3
