#!/bin/bash
# rocprofv3 kernel trace of the default bench command; summaries land in gpurun_out/prof_<tag>/
tag=${1:-r01}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_$tag/bench_stdout.log 2>&1
find gpurun_out/prof_$tag -name "*stats*" | head
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
echo "== $f"; cat "$f"
# keep the merge small: drop the per-dispatch trace
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -size +8M -delete
