#!/bin/bash
# round 6: the last run on the tree as committed -- smoke, the whole GPU tier, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_last
mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -x -s 2>&1 | grep -v "amdgpu.ids" > $O/gpu_tests.log; tail -2 $O/gpu_tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<PY
import json
j=json.load(open("$O/bench.json"))
print(j["ms_per_step"], j["value"], "traffic", j["roofline"]["traffic"], "machine probes", j["roofline"]["machine"]["probes"], j["build_id"])
print("cpu", j["cpu_baseline"]["value"], "regions", j["kernel_region_host"]["ms_per_call"], j["kernel_region_host_packed"]["ms_per_call"], "e2e", j["end_to_end"]["value"], j["end_to_end"]["median_value"])
PY
