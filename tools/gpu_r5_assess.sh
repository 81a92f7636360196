#!/bin/bash
# k_assess at 4 / 6 / 8 waves per SIMD (register cap 128 / 80 / 64): -s 1 and counting side lines
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in libntedit_hip.so libntedit_hip_a6.so libntedit_hip_a8.so; do
echo "== $lib"
export NTEDIT_HIP_LIB=$PWD/ntedit_amd/$lib
python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 4 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('snv', d['value'], d['ms_per_step'], d['phases_ms'])"
python bench.py --counting --bases 250e6 --contig-len 100000 --steps 4 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('counting', d['value'], d['ms_per_step'], d['phases_ms'])"
done
unset NTEDIT_HIP_LIB
bash tools/pmc_assess.sh
