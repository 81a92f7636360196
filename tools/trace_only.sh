#!/bin/bash
# kernel-trace stats of one bench invocation: tools/trace_only.sh <tag> [bench args]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o trace -- python $ROOT/bench.py "$@" > $OUT/bench.json 2> $OUT/err.txt
find $OUT/t -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/t
python $ROOT/tools/kstats.py $OUT/kernel_stats.csv
