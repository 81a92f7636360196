#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_l
mkdir -p $O
for S in iid genome; do for V in 0 1; do
  NTEDIT_HIP_DEBUG=1 python bench.py --structure $S --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune sort_events=$V > $O/b_${S}_$V.json 2> $O/b_${S}_$V.err
  python -c "
import json; j=json.load(open('$O/b_${S}_$V.json')); print('$S sort_events=$V', j['ms_per_step'], j['phases_ms'], j['events'])"
  grep -E "events [0-9]+ \(round" $O/b_${S}_$V.err | tail -1 | cut -c60-300
done; done
