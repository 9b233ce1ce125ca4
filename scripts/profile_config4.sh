#!/bin/bash
# rocprofv3 kernel stats of the bench workload through the double array (BASELINE.json configs[3]: the same bytes as
# configs[1], tokenizer_de.datok) -> gpurun_out/r02/config4_*; copy the two files into profiles/ by hand
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02/prof_c4 -o bench -- python3 bench.py --model tests/golden/models/tokenizer_de.datok --no-cpu-baseline > gpurun_out/r02/config4_bench_line_under_rocprof.json 2> gpurun_out/r02/config4_stderr.log || exit 1
f=$(find gpurun_out/r02/prof_c4 -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r02/config4_kernel_stats.csv
find gpurun_out/r02/prof_c4 -name "*kernel_trace.csv" -size +8M -delete
head -5 gpurun_out/r02/config4_kernel_stats.csv | cut -c1-160
tail -c 400 gpurun_out/r02/config4_bench_line_under_rocprof.json
