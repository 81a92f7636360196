#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_o
mkdir -p $O
for S in iid genome; do
  NTEDIT_HIP_DEBUG=1 python bench.py --structure $S --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather > $O/b_${S}.json 2> $O/b_${S}.err
  python -c "
import json; j=json.load(open('$O/b_${S}.json')); print('$S', j['ms_per_step'], j['phases_ms'])"
  grep -E "events [0-9]+ \(round" $O/b_${S}.err | tail -1 | cut -c60-300
done
python bench.py --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > $O/c2.json 2>/dev/null
python -c "
import json; j=json.load(open('$O/c2.json')); print('configs2', j['ms_per_step'], j['phases_ms'])"
python bench.py --bases 3.75e8 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > $O/s.json 2>/dev/null
python -c "
import json; j=json.load(open('$O/s.json')); print('375 Mbp', j['ms_per_step'], j['phases_ms'])"
