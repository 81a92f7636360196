#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_k
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "snv or counting or cbf or polish_matches or sweep_rich" > $O/tests.log 2>&1
grep -E "^\[|passed|failed|Error|assert" $O/tests.log | cut -c1-400 | tail -8
for M in 0 1; do
python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather --tune candmap=$M > $O/bench_snv_m$M.json 2>$O/bench_snv_m$M.err
python -c "
import json; j=json.load(open('$O/bench_snv_m$M.json')); print('candmap=$M', j['ms_per_step'], j['value'], j['phases_ms'])"
done
python bench.py --counting --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > $O/bench_counting.json 2>/dev/null
python -c "
import json; j=json.load(open('$O/bench_counting.json')); print('counting', j['ms_per_step'], j['value'], j['phases_ms'])"
