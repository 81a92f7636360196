#!/bin/bash
# per-GPU rate at the shard sizes of a 3 Gbp strong-scaling run (N = 8, 4, 2, 1); a context's FIRST call after
# ntedit_hip_reserve ("cold") next to the warm rate; configs[2] warm / cold / cold without reserve
cd "$GRAFT_REPO_ROOT" || exit 1
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], d['phases_ms'], d['roofline']['pipeline']['partition_ms'], d['roofline']['pipeline']['probe_ms'], 'reserve_s', d.get('reserve_s'))"; }
B="--no-regions --no-cpu-baseline --no-gather ${BENCH_ARGS:-}"
for b in 3.75e8 7.5e8 1.5e9 3e9; do
python bench.py --bases $b --steps 5 --warmup 2 $B 2>/dev/null | line "$b warm"
python bench.py --bases $b --steps 1 --warmup 0 $B 2>/dev/null | line "$b first-call-after-reserve"
done
python bench.py --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 $B 2>/dev/null | line "configs[2] warm"
python bench.py --bases 250e6 --contig-len 100000 --steps 1 --warmup 0 $B 2>/dev/null | line "configs[2] first-call-after-reserve"
python bench.py --bases 250e6 --contig-len 100000 --steps 1 --warmup 0 --no-reserve $B 2>/dev/null | line "configs[2] first-call-no-reserve"
