#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_g
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pieces or packed or chunk_pipeline" > $O/tests.log 2>&1
tail -3 $O/tests.log
for PB in 33554432 67108864; do
timeout 1200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --no-gather --tune h2d_piece=$PB > $O/bench_regions_$PB.json 2> $O/bench_regions_$PB.err
python - <<PY
import json
j=json.load(open("$O/bench_regions_$PB.json"))
for k in j:
    if "region" in k or "host" in k:
        print("piece $PB:", k, j[k]["ms_per_call"], j[k]["value"], j[k]["screen_ms_incl_copy_waits"], j[k]["screen_launches"])
PY
done
