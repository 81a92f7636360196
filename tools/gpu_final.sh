#!/bin/bash
# Round-end record of the shipped build on the GPU box: the whole GPU test tier, the default bench line (regions and CPU
# baseline included), the kernel-trace stats of the bench workload, and a fuzz run.   tools/gpu_final.sh <tag> [fuzz minutes]
TAG=${1:-final}; FZ=${2:-4}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu -x -s 2>&1 | grep -v "amdgpu.ids" > $OUT/gpu_tests.log; tail -2 $OUT/gpu_tests.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench.json
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -o trace -- \
   python $GRAFT_REPO_ROOT/bench.py --bases 3e9 --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-regions --no-reserve > $GRAFT_REPO_ROOT/$OUT/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$OUT/trace.err)
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; ; rm -rf $OUT/trace
head -8 $OUT/kernel_stats.csv | cut -c1-160
timeout $((FZ*60+200)) python tests/tools/fuzz_parity.py --gpu --minutes $FZ --seed 424242 2>&1 | tail -3 > $OUT/fuzz.log; cat $OUT/fuzz.log
