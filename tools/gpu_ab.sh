#!/bin/bash
# A/B of library variants on the GPU box: tools/gpu_ab.sh <outdir> libA.so libB.so ...  (files under ntedit_amd/)
# per variant: the binned-screening parity check (guarded by per-case timeouts), then the 3 Gbp screening timed per stage.
# BENCH_ARGS: extra bench.py arguments (e.g. "--tune records_uncached=1"); SKIP_CHECK=1: timing only
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$1; shift
mkdir -p "$OUT"
for lib in "$@"; do
  echo "== $lib ${BENCH_ARGS:-}"
  export NTEDIT_HIP_LIB=$PWD/ntedit_amd/$lib
  if [ -n "${SKIP_CHECK:-}" ] || timeout 300 python tests/tools/screen_check.py --timeout 45 > "$OUT/check_$lib.log" 2>&1; then
    timeout 240 python bench.py --screen-only --steps 2 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune bin_timing=1 ${BENCH_ARGS:-} 2>&1 | grep -E "binned chunk|scatter:" | tail -2 | cut -c1-400
  else
    tail -5 "$OUT/check_$lib.log"
  fi
done
