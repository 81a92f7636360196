#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_gpu_multigpu.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3
FUZZ_TRACE=1 timeout 900 python tests/tools/fuzz_parity.py --gpu --seed 424242 --iters 80 > gpurun_out/r5_fuzzdbg.log 2>&1
grep -B1 "MISMATCH" gpurun_out/r5_fuzzdbg.log | cut -c1-900
tail -1 gpurun_out/r5_fuzzdbg.log
