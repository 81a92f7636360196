#!/bin/bash
# round 6: contig-end window -- parity subset, then the bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_b
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not every_contig and not deep_search and not reserve_first" > $O/tests.log 2>&1
tail -5 $O/tests.log
for B in 3.0e9 1.0e9; do
  for S in genome iid; do
    NTEDIT_HIP_DEBUG=1 timeout 900 python bench.py --structure $S --bases $B --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather \
       > $O/bench_${S}_${B}.json 2> $O/bench_${S}_${B}.err
    echo "== $S $B"; python -c "
import json; j=json.load(open('$O/bench_${S}_${B}.json')); print(j['ms_per_step'], j['value'], j['phases_ms'], j['events'])"
    grep -E "events [0-9]+ \(round" $O/bench_${S}_${B}.err | tail -1 | cut -c1-400
  done
done
