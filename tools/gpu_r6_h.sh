#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_h
mkdir -p $O
for F in 0 3 6 12; do
python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather --tune defer_fail_snv=$F > $O/bench_snv_$F.json 2>/dev/null
python -c "
import json; j=json.load(open('$O/bench_snv_$F.json')); print('defer_fail_snv=$F', j['ms_per_step'], j['value'], j['phases_ms'], j['events'])"
done
NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so NTEDIT_HIP_DEBUG=1 python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 1 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>&1 >/dev/null | grep -E "per-event launches|events [0-9]+ \(round" | tail -3 | cut -c1-900
