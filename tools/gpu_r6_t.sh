#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_t
mkdir -p $O
for T in "inline_tries=2" "inline_tries=4" "inline_tries=6" "inline_tries=8" "inline_tries=12" "inline_tries=16"; do
  for S in iid genome; do
  python bench.py --structure $S --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune $T > $O/b.json 2>/dev/null
  python -c "
import json; j=json.load(open('$O/b.json')); print('$T $S', j['ms_per_step'], j['phases_ms']['machine_launches_sum'])"
  done
done
