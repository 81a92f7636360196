#!/bin/bash
# event start grid (extra event starts inside long absent runs) against the tail of the machine launches
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 3.75e8 3e9; do for g in 256 64 32 16; do
python bench.py --bases $b --start-grid $g --steps 4 --warmup 2 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b grid $g', d['value'], d['ms_per_step'], d['phases_ms'], d['events']['event_starts'], d['events']['deferred_to_sweep_pass'])"
done; done
