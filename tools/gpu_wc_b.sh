#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for dbg in 0 256; do
echo "== WC debug=$dbg"
NTEDIT_HIP_MACHINE_DEBUG=$dbg NTEDIT_HIP_BIN_WC=1 NTEDIT_HIP_BIN_TIMING=1 python bench.py --steps 1 --warmup 1 --bases 1.0e9 --screen-only --screen-mode 2 --no-gather --no-cpu-baseline 2>&1 | grep -E "binned chunk|wc partition" | tail -2 | cut -c1-250
done
