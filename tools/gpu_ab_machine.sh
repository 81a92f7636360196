#!/bin/bash
# A/B of library variants on the GPU box, whole step: tools/gpu_ab_machine.sh libA.so libB.so ...  (files under ntedit_amd/)
# per variant: bench.py's 3 Gbp step (HBM-resident) with its per-phase sums; BENCH_ARGS adds bench.py arguments
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "$@"; do
  export NTEDIT_HIP_LIB=$PWD/ntedit_amd/$lib
  timeout 300 python bench.py --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather ${BENCH_ARGS:-} 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$lib', d['ms_per_step'], d['phases_ms'])"
done
