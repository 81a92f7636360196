#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binned" 2>&1 | tail -15 > gpurun_out/wc_tests.log
tail -3 gpurun_out/wc_tests.log
for wgs in 256; do
echo "== WC wgs=$wgs"
NTEDIT_HIP_WC_WGS=$wgs NTEDIT_HIP_BIN_WC=1 NTEDIT_HIP_BIN_TIMING=1 python bench.py --steps 2 --warmup 1 --screen-only --screen-mode 2 --no-gather --no-cpu-baseline 2>&1 | grep -E "binned chunk|metric|wc partition" | tail -7 | cut -c1-250
done
