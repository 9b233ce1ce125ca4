#!/bin/bash
# Everything profiles/ holds for a round: the plain bench line, the bench line under rocprofv3 with its
# kernel stats, and the FETCH_SIZE / WRITE_SIZE passes (counters only, kernel trace only).
tag=${1:-r01}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
python3 bench.py > gpurun_out/$tag/bench_line.json 2> gpurun_out/$tag/bench_stderr.log || exit 1
tail -c 600 gpurun_out/$tag/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof -o bench -- python3 bench.py --no-cpu-baseline --no-e2e > gpurun_out/$tag/bench_line_under_rocprof.json 2> gpurun_out/$tag/prof_stderr.log || exit 1
f=$(find gpurun_out/$tag/prof -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/$tag/kernel_stats.csv; cat gpurun_out/$tag/kernel_stats.csv
find gpurun_out/$tag/prof -name "*kernel_trace.csv" -size +8M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 150 bash scripts/pmc_bench.sh ${tag}_$c $c > gpurun_out/$tag/pmc_$c.txt 2>&1 || exit 1
done
cat gpurun_out/$tag/pmc_FETCH_SIZE.txt gpurun_out/$tag/pmc_WRITE_SIZE.txt > gpurun_out/$tag/pmc_traffic.txt
grep -A1 "k_spec_both" gpurun_out/$tag/pmc_traffic.txt
