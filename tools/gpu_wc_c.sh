#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binned or every_contig or deep_search" -s 2>&1 | grep -E "passed|failed|configs|Error" | cut -c1-330 > gpurun_out/wc_tests2.log
cat gpurun_out/wc_tests2.log
