#!/bin/bash
# round 6, after the last source change (packed pieces of 128 MiB): the records that are stamped with the build id -- the
# default bench line, kernel-trace stats, machine probe counts, PMC passes -- and the packed / arriving-batch tests again
cd "$GRAFT_REPO_ROOT" || exit 1
T=r6final2
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pieces or packed or chunk_pipeline or reserve" > gpurun_out/$T/tests_subset.log 2>&1; tail -2 gpurun_out/$T/tests_subset.log
timeout 900 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; cut -c1-300 gpurun_out/$T/bench.json
bash tools/gpu_machine_probes.sh > gpurun_out/$T/machine_probes.log 2>&1
cp gpurun_out/machine_probes.json gpurun_out/$T/
bash tools/profile_gpu.sh r6b > gpurun_out/$T/profile_gpu.log 2>&1
timeout 900 python bench.py > gpurun_out/$T/bench_with_traffic.json 2> gpurun_out/$T/bench2.err
python - <<PY
import json
j=json.load(open("gpurun_out/$T/bench_with_traffic.json"))
print(j["ms_per_step"], j["roofline"]["traffic"], j["roofline"].get("traffic_source"), j["roofline"]["machine"])
for k in ("kernel_region_host","kernel_region_host_packed"):
    print(k, j[k]["ms_per_call"], j[k]["value"])
print("e2e", j["end_to_end"]["value"], j["end_to_end"]["median_value"])
PY
