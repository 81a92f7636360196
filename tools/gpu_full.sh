#!/bin/bash
# the round-end sequence, as the driver runs it: the whole GPU test tier, smoke, the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/full_tests.log; cat gpurun_out/full_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/full_bench.json 2> gpurun_out/full_bench.err; cat gpurun_out/full_bench.json | cut -c1-6000
