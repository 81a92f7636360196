#!/bin/bash
# round 4: phase timers + duration histogram of the wavefront-per-event kernel (profile build)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r4d; mkdir -p $OUT
export NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so
for v in "$@"; do
  t=""; for kv in $v; do t="$t --tune $kv"; done
  echo "== $v"
  NTEDIT_HIP_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-regions --no-cpu-baseline --no-gather $t ${BENCH_ARGS:-} 2> $OUT/err.log |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['phases_ms'], d.get('events'))"
  grep -E "chunk 1/1|wave-kernel|inside failing" $OUT/err.log | tail -4 | cut -c1-900
done
