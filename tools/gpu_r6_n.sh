#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_n
mkdir -p $O
for S in iid genome; do for V in 0 1; do
  NTEDIT_HIP_DEBUG=1 python bench.py --structure $S --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune lockstep=$V > $O/b_${S}_$V.json 2> $O/b_${S}_$V.err
  python -c "
import json; j=json.load(open('$O/b_${S}_$V.json')); print('$S lockstep=$V', j['ms_per_step'], j['phases_ms'])"
  grep -E "events [0-9]+ \(round" $O/b_${S}_$V.err | tail -1 | cut -c60-300
done; done
for V in 0 1; do
python bench.py --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather --tune lockstep=$V > $O/c2_$V.json 2>/dev/null
python -c "
import json; j=json.load(open('$O/c2_$V.json')); print('configs2 lockstep=$V', j['ms_per_step'], j['phases_ms'])"
python bench.py --bases 3.75e8 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather --tune lockstep=$V > $O/s_$V.json 2>/dev/null
python -c "
import json; j=json.load(open('$O/s_$V.json')); print('375 Mbp lockstep=$V', j['ms_per_step'], j['phases_ms'])"
done
