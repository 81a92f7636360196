#!/bin/bash
# round 6: side lines after the window / defer_fail / assess changes, + device tables test
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_d
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -s -k "snv or counting or device_tables or contig_ends or golden or cbf" > $O/tests.log 2>&1
grep -E "^\[|passed|failed" $O/tests.log | cut -c1-300
python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > $O/bench_snv_250Mbp.json 2>$O/bench_snv.err
python bench.py --counting --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > $O/bench_counting_250Mbp.json 2>$O/bench_counting.err
for f in snv counting; do python -c "
import json; j=json.load(open('$O/bench_${f}_250Mbp.json')); print('$f', j['ms_per_step'], j['value'], j['phases_ms'], j['events'])"; done
NTEDIT_HIP_DEBUG=1 python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 1 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>&1 >/dev/null | grep -E "events [0-9]+ \(round" | tail -1 | cut -c1-300
