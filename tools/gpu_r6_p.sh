#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_p
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "polish_matches or contig_ends or sweep_rich or golden or genome_like_cases or demo" > $O/tests.log 2>&1; tail -2 $O/tests.log
for S in iid genome; do
  NTEDIT_HIP_DEBUG=1 python bench.py --structure $S --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather > $O/b_${S}.json 2> $O/b_${S}.err
  python -c "
import json; j=json.load(open('$O/b_${S}.json')); print('$S', j['ms_per_step'], j['phases_ms'])"
  grep -E "events [0-9]+ \(round" $O/b_${S}.err | tail -1 | cut -c60-300
done
python bench.py --counting --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > $O/cnt.json 2>/dev/null
python -c "
import json; j=json.load(open('$O/cnt.json')); print('counting', j['ms_per_step'], j['phases_ms'])"
timeout 400 python tests/tools/fuzz_parity.py --gpu --minutes 5 --seed 77 2>&1 | tail -1
