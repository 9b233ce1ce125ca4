#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r02/t_v2.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/r02/t_v2.log
for v in A B; do echo "== lib$v"; DATOK_GPU_LIB=$PWD/ab/lib$v.so timeout -k 10 200 python scripts/sweep2.py 64,128 1,3 2>&1 | tail -3; done
for v in B K1 K3 K8; do echo "== lib$v (compaction skipped)"; DATOK_EXP_SKIP=8 DATOK_GPU_LIB=$PWD/ab/lib$v.so timeout -k 10 200 python scripts/sweep2.py 128 1,3 2>&1 | tail -2; done
timeout -k 10 300 python scripts/robust.py tokenizer_de.matok > gpurun_out/r02/robust.log 2>&1; echo "robust rc=$?"; cat gpurun_out/r02/robust.log
