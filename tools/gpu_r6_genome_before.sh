#!/bin/bash
# round 6, step (b) of VERDICT r5 item 1: the SHIPPED build (round 5's kernels) on the genome-like workload, before any fix.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_genome_before
mkdir -p $O
for B in 3.0e9 1.0e9; do
  for S in iid genome; do
    NTEDIT_HIP_DEBUG=1 timeout 900 python bench.py --structure $S --bases $B --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather \
       --tune bin_timing=1 > $O/bench_${S}_${B}.json 2> $O/bench_${S}_${B}.err
    echo "== $S $B"; cut -c1-300 $O/bench_${S}_${B}.json
    grep -E "binned chunk|events [0-9]+ \(round" $O/bench_${S}_${B}.err | tail -4
  done
done
