#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for dbg in 0 512 1024 2048 4096 8192 12288; do
echo -n "debug=$dbg: "
NTEDIT_HIP_BIN_TIMING=1 NTEDIT_HIP_MACHINE_DEBUG=$dbg timeout 100 python bench.py --steps 1 --warmup 1 --bases 1.0e9 --screen-only --no-gather --no-cpu-baseline 2>&1 | grep "binned chunk" | tail -1 | sed 's/.*bits: //'
done
