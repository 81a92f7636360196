#!/bin/bash
# one PMC pass over a bench invocation: tools/pmc_once.sh <tag> "<counters>" [bench args]
TAG=$1; PMC=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d $OUT/t -o pmc -- python $ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/err.txt
f=$(find $OUT/t -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get("Kernel_Name", "?")[:48]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r.get("Dispatch_Id"))
for k in agg:
    if 'nte::' in k: print(k, "dispatches=%d" % len(cnt[k]), " ".join("%s=%.5g" % kv for kv in sorted(agg[k].items())))
PY
rm -rf $OUT/t; tail -3 $OUT/err.txt
