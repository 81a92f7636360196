#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_j
mkdir -p $O
python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 2 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune candmap=1 --tune bin_timing=1 > $O/b.json 2>$O/b.err
grep "binned chunk" $O/b.err | tail -3 | cut -c1-400
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -o trace -- \
   python $GRAFT_REPO_ROOT/bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune candmap=1 --no-reserve > /dev/null 2> $GRAFT_REPO_ROOT/$O/trace.err)
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/trace
head -12 $O/kernel_stats.csv | cut -c1-200
