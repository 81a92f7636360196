#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binned" 2>&1 | tail -15 > gpurun_out/wc_tests.log
tail -3 gpurun_out/wc_tests.log
NTEDIT_HIP_BIN_TIMING=1 python bench.py --steps 2 --warmup 1 --screen-only --no-gather --no-cpu-baseline 2>&1 | grep -E "binned chunk|metric|wc partition" | tail -4 | cut -c1-250
