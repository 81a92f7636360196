#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_m
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "cli or demo or golden or multigpu or polish_matches or many" > $O/tests.log 2>&1; tail -3 $O/tests.log
for i in 1 2; do
timeout 900 python bench.py --no-cpu-baseline --no-gather > $O/bench_$i.json 2> $O/bench_$i.err
python - <<PY
import json
j=json.load(open("$O/bench_$i.json"))
e=j["end_to_end"]; print("e2e", e["value"], e["median_value"], e["region_s_all_runs"], e["stage_s_all_runs"][0])
PY
done
NTEDIT_RENDER_ONE_WRITER=1 timeout 900 python bench.py --no-cpu-baseline --no-gather > $O/bench_one.json 2> $O/bench_one.err
python - <<PY
import json
j=json.load(open("$O/bench_one.json"))
e=j["end_to_end"]; print("one writer: e2e", e["value"], e["median_value"], e["region_s_all_runs"], e["stage_s_all_runs"][0])
PY
