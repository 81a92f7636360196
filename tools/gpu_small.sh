#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 3.75e8 7.5e8 1.5e9; do
python bench.py --bases $b --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b', d['value'], d['ms_per_step'], d['phases_ms'], d['roofline']['pipeline']['partition_ms'], d['roofline']['pipeline']['probe_ms'], d['roofline']['launches_per_step'])"
done
