#!/bin/bash
# round 6: defer_fail sweep on the genome-like and i.i.d. 3 Gbp drafts
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_c
mkdir -p $O
for S in genome iid; do
  for F in 0 2 3 4 6 8; do
    NTEDIT_HIP_DEBUG=1 timeout 900 python bench.py --structure $S --bases 3e9 --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune defer_fail=$F \
       > $O/bench_${S}_$F.json 2> $O/bench_${S}_$F.err
    echo "== $S defer_fail=$F"; python -c "
import json; j=json.load(open('$O/bench_${S}_$F.json')); print(j['ms_per_step'], j['phases_ms'], j['events'])"
    grep -E "events [0-9]+ \(round" $O/bench_${S}_$F.err | tail -1 | cut -c60-400
  done
done
