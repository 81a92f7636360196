#!/bin/bash
# per-GPU rate at the shard sizes of a 3 Gbp strong-scaling run (N = 8, 4, 2), configs[2] warm and cold
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 3.75e8 7.5e8 1.5e9 3e9; do
python bench.py --bases $b --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b', d['value'], d['ms_per_step'], d['phases_ms'], d['roofline']['pipeline']['partition_ms'], d['roofline']['pipeline']['probe_ms'], d['roofline']['launches_per_step'])"
done
echo "configs[2] warm (5 steps after 2) / cold (1 step, no warm-up)"
python bench.py --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('warm', d['value'], d['ms_per_step'], d['phases_ms'])"
python bench.py --bases 250e6 --contig-len 100000 --steps 1 --warmup 0 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('cold', d['value'], d['ms_per_step'], d['phases_ms'])"
