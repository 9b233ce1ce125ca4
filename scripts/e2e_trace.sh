#!/bin/bash
# timeline of one pipeline run with results coming to the host: kernels and copies (rocprofv3, no counters)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_e2e
E2E_ONLY=1 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_e2e -- python3 scripts/e2e_diag2.py 2>&1 | tail -3
find gpurun_out/prof_e2e -name "*.csv" | head; 
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob('gpurun_out/prof_e2e/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Direction'], int(r.get('Bytes', r.get('Size', 0)) or 0)))
for f in glob.glob('gpurun_out/prof_e2e/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K:' + r['Kernel_Name'][:22], 0))
rows.sort()
t0 = rows[0][0]
# the last 140 events: steady state of the last run
out = open('gpurun_out/e2e_timeline.txt', 'w')
for s, e, what, n in rows[-220:]:
    out.write("%10.3f %10.3f  %8.3f ms  %-28s %d\n" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, what, n))
out.close()
print(len(rows), 'events')
PY
