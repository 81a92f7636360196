#!/bin/bash
# round 6: the whole GPU test tier
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r6_full}
mkdir -p $O
timeout 5400 python -m pytest tests -m gpu -x -q -s > $O/gpu_tests.log 2>&1
grep -E "^\[|passed|failed|error" $O/gpu_tests.log | cut -c1-400 | tail -30
