#!/bin/bash
# round 5, first check of: interleaved seed tables, word-wise code reads, slice sweep in the probe stage, ntedit_hip_reserve
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu -k "not every_contig and not full_size" > gpurun_out/r5a_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5a_tests.log
B="--steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline']['pipeline']
print('$1', 'value', d['value'], 'ms', d['ms_per_step'], d['phases_ms'], 'partition', p['partition_ms'], 'probe', p['probe_ms'], 'reserve_s', d.get('reserve_s'))"; }
{
python bench.py $B 2>/dev/null | line "3Gbp sweep"
python bench.py $B --tune probe_sweep=0 2>/dev/null | line "3Gbp nosweep"
python bench.py $B --bases 3.75e8 2>/dev/null | line "375Mbp sweep"
python bench.py $B --bases 3.75e8 --tune probe_sweep=0 2>/dev/null | line "375Mbp nosweep"
python bench.py $B --bases 7.5e8 2>/dev/null | line "750Mbp sweep"
python bench.py --bases 250e6 --contig-len 100000 --steps 1 --warmup 0 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | line "configs2 cold after reserve"
python bench.py --bases 250e6 --contig-len 100000 --steps 1 --warmup 0 --no-regions --no-cpu-baseline --no-gather --no-reserve 2>/dev/null | line "configs2 cold no reserve"
python bench.py --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | line "configs2 warm"
python bench.py --bases 3.75e8 --steps 1 --warmup 0 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | line "375Mbp cold after reserve"
} > gpurun_out/r5a_bench.txt 2>&1
