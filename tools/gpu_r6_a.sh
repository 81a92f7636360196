#!/bin/bash
# round 6: first run of the overflow-list rework -- the binned screening tests, then the genome-like bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_a
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binned or chunk_pipeline or pieces or screen_bitmap or edge_sizes or reserve" > $O/tests.log 2>&1
tail -5 $O/tests.log
for B in 3.0e9 1.0e9; do
  for S in genome iid; do
    NTEDIT_HIP_DEBUG=1 timeout 900 python bench.py --structure $S --bases $B --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather \
       --tune bin_timing=1 > $O/bench_${S}_${B}.json 2> $O/bench_${S}_${B}.err
    echo "== $S $B"; cut -c1-200 $O/bench_${S}_${B}.json
    grep -E "binned chunk|events [0-9]+ \(round|direct kernel" $O/bench_${S}_${B}.err | tail -3 | cut -c1-400
  done
done
