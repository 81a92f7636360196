#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_r
mkdir -p $O
B="--no-regions --no-cpu-baseline --no-gather"
for V in _as5 _as6 _as7; do
  NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip$V.so python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 5 --warmup 2 $B > $O/snv$V.json 2>/dev/null
  NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip$V.so python bench.py --counting --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 $B > $O/cnt$V.json 2>/dev/null
  python -c "
import json; j=json.load(open('$O/snv$V.json')); print('assess blocks $V snv', j['ms_per_step'], j['phases_ms']); j=json.load(open('$O/cnt$V.json')); print('   counting', j['ms_per_step'], j['phases_ms'])"
done
