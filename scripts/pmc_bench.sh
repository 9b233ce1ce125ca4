#!/bin/bash
# rocprofv3 PMC pass (counters only, kernel-trace only) over a short bench run.
# usage: pmc_bench.sh <tag> "<counter list>" [bench args]
tag=$1; ctrs=$2; shift 2
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag
mkdir -p $out
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out -o pmc -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-e2e --parity-docs 16 "$@" > $out/stdout.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
f=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"].split("(")[0][:60]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,v in agg.items():
    print(k)
    for c,val in v.items(): print("   %-28s %16.1f per-dispatch" % (c, val/max(1,n[(k,c)])))
PY
find $out -name "*.csv" -size +4M -delete
