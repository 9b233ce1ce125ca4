#!/bin/bash
cd /tmp
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r02
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02/prof_rich -o rich -- python $R/scripts/prof_rich.py > $R/gpurun_out/r02/prof_rich.log 2>&1; echo "rich rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r02/prof_b1 -o b1 -- python $R/bench.py --streams 1 --steps 30 --warmup 3 --no-cpu-baseline --parity-docs 0 > $R/gpurun_out/r02/prof_b1.log 2>&1; echo "b1 rc=$?"
ls $R/gpurun_out/r02/prof_rich $R/gpurun_out/r02/prof_b1
