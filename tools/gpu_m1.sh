#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multigpu.py -x -q -m gpu -k "not every_contig and not deep_search" 2>&1 | tail -3
NTEDIT_HIP_DEBUG=1 python bench.py --steps 2 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>&1 | grep -E "round A|metric" | tail -2 | cut -c1-420
timeout 300 python tests/tools/fuzz_parity.py --gpu --minutes 3 --seed 4321 2>&1 | tail -3
