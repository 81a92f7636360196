#!/bin/bash
# round 4: library variants of the machine kernels on the 3 Gbp step (+ phase timers of the wavefront-per-event kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r4b; mkdir -p $OUT
for lib in "$@"; do
  export NTEDIT_HIP_LIB=$PWD/ntedit_amd/$lib
  echo "== $lib"
  NTEDIT_HIP_DEBUG=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather ${BENCH_ARGS:-} 2> $OUT/err_$lib.log |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['phases_ms'], d.get('events'))"
  grep -E "chunk 1/1|wave-kernel|inside failing" $OUT/err_$lib.log | tail -3 | cut -c1-700
done
