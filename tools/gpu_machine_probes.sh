#!/bin/bash
# Filter bytes gathered by the event-machine launches of one step (the edit search's probes), counted lane by
# lane by the profile build; appends a record to gpurun_out/machine_probes.json (merge it into profiles/machine_probes.json).
#   tools/gpu_machine_probes.sh [bench args...]       (run `make -C ntedit_amd/csrc profile` first)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so
NTEDIT_HIP_DEBUG=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-regions --no-cpu-baseline --no-gather "$@" > gpurun_out/mp_bench.json 2> gpurun_out/mp_err.log
python - <<'PY'
import json, re, os
line = json.loads(open("gpurun_out/mp_bench.json").readline())
g = [l for l in open("gpurun_out/mp_err.log") if "machine filter gathers" in l]
m = re.search(r"thread-per-event launches (\d+), wavefront-per-event launches (\d+)", g[-1])
cfg = line["config"]
rec = {k: cfg[k] for k in ("workload_bytes", "k", "hashes", "filter_bytes", "snv", "counting")}
rec.update(build_id=line["build_id"], gathers_thread_launches=int(m.group(1)), gathers_wave_launches=int(m.group(2)),
           events=line.get("events"), workload=cfg["workload"])
path = "gpurun_out/machine_probes.json"
doc = json.load(open(path)) if os.path.exists(path) else {
    "source": "tools/gpu_machine_probes.sh: bench.py on the profile build (make profile), one step; every lane counts the filter bytes it gathers",
    "records": []}
doc["records"].append(rec)
json.dump(doc, open(path, "w"), indent=1)
print(rec)
PY
