#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_e
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "genome_like_cases" > $O/tests.log 2>&1
grep -E "^\[|passed|failed|Error|error" $O/tests.log | cut -c1-600
timeout 900 python tests/tools/fuzz_genome_like.py --minutes 8 --seed 6161 > $O/fuzz_genome_like.log 2>&1
tail -25 $O/fuzz_genome_like.log | cut -c1-600
