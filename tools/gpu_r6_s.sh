#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_s
mkdir -p $O
for G in 64 128 512 1024; do
  for S in genome iid; do
  python bench.py --structure $S --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --start-grid $G > $O/b_${G}_$S.json 2>/dev/null
  python -c "
import json; j=json.load(open('$O/b_${G}_$S.json')); print('grid $G $S', j['ms_per_step'], j['phases_ms'], j['events']['event_starts'])"
  done
done
