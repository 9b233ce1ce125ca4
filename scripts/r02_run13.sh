#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t_small.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r02/t_small.log
DATOK_SMALL_MAX=100000 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_exact_and_replay.py -x -q -m gpu -k "not config5 and not zipf_full and not large" > gpurun_out/r02/t_small2.log 2>&1; echo "tests (everything one lane per document) rc=$?"
tail -5 gpurun_out/r02/t_small2.log
timeout -k 10 200 python scripts/tiny_docs.py 2>&1 | tail -3
ONLY=3 timeout -k 10 200 python scripts/configs.py 2>&1 | tail -1
timeout -k 10 100 python scripts/sweep2.py 128 1,3 2>&1 | tail -1
timeout -k 10 120 python scripts/soak.py 12 4000 2>&1 | tail -1
