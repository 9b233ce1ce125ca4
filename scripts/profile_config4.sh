#!/bin/bash
# rocprofv3 kernel stats of the bench workload through the double array (BASELINE.json configs[3]: the same bytes as
# configs[1], tokenizer_de.datok), dense layout (the default) and the file's {base, check} pairs (test hook NO_DENSE)
# -> gpurun_out/<tag>/config4_{dense,pairs}_*; copy into profiles/ by hand.   usage: profile_config4.sh [tag]
tag=${1:-r03}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
for layout in dense pairs; do
  if [ $layout = pairs ]; then export DATOK_NO_DENSE=1; else unset DATOK_NO_DENSE; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/prof_c4_$layout -o bench -- python3 bench.py --model tests/golden/models/tokenizer_de.datok --no-cpu-baseline --no-e2e > gpurun_out/$tag/config4_${layout}_bench_line_under_rocprof.json 2> gpurun_out/$tag/config4_${layout}_stderr.log || exit 1
  f=$(find gpurun_out/$tag/prof_c4_$layout -name "*kernel_stats.csv" | head -1)
  cp "$f" gpurun_out/$tag/config4_${layout}_kernel_stats.csv
  find gpurun_out/$tag/prof_c4_$layout -name "*kernel_trace.csv" -size +8M -delete
  head -4 gpurun_out/$tag/config4_${layout}_kernel_stats.csv | cut -c1-160
  python3 -c "
import json,sys
for l in open('gpurun_out/$tag/config4_${layout}_bench_line_under_rocprof.json'):
    if l.startswith('{'):
        j=json.loads(l); print('$layout', j['value'], 'MB/s; one batch', j['streams_1']['value'], '|', j['config']['table_layout'], '|', j['roofline']['kernel'])
"
done
