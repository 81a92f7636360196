#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_f
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pieces or packed or chunk_pipeline" > $O/tests.log 2>&1
tail -3 $O/tests.log
timeout 1200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e > $O/bench_regions.json 2> $O/bench_regions.err
python - <<PY
import json
j=json.load(open("$O/bench_regions.json"))
print(j["ms_per_step"], j["value"])
for k in j:
    if "region" in k or "host" in k:
        print(k, json.dumps(j[k])[:900])
PY
timeout 1200 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-e2e --tune h2d_fixed_schedule=1 > $O/bench_regions_fixed.json 2> $O/bench_regions_fixed.err
python - <<PY
import json
j=json.load(open("$O/bench_regions_fixed.json"))
for k in j:
    if "region" in k or "host" in k:
        print("fixed schedule:", k, json.dumps(j[k])[:900])
PY
for W in ; do
python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather --tune snv_wave=$W > $O/bench_snv_w$W.json 2>/dev/null
python -c "
import json; j=json.load(open('$O/bench_snv_w$W.json')); print('snv_wave=$W', j['ms_per_step'], j['value'], j['phases_ms'])"
done
