#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp; mkdir -p gpurun_out/r02
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "speculative or edge or rich or longer or long or golden or config2" > gpurun_out/r02/t_tag.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r02/t_tag.log
timeout -k 10 200 python scripts/robust.py 2>&1 | tail -4
