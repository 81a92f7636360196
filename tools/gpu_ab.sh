#!/bin/bash
# A/B of library variants: tools/gpu_ab.sh libA.so libB.so ... (files under ntedit_amd/)
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in "$@"; do
echo "== $lib"
NTEDIT_HIP_LIB=$PWD/ntedit_amd/$lib NTEDIT_HIP_DEBUG=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>&1 | grep -E "machine .* ms|metric" | tail -2 | sed 's/.*sweeps/sweeps/' | cut -c1-200
done
