#!/bin/bash
# defer_walk: an event of the thread-per-event launch that has walked N positions goes to the wavefront-per-event launch
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "polish_matches_oracle or binned_screen" 2>&1 | grep -v amdgpu | tail -2
for w in 0 8 24 64 200; do
for b in 3.75e8 3e9; do
NTEDIT_HIP_DEBUG=1 python bench.py --bases $b --steps 3 --warmup 2 --no-regions --no-cpu-baseline --no-gather --tune defer_walk=$w 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('defer_walk=$w bases=$b', 'value', d['value'], 'ms', d['ms_per_step'], d['phases_ms'], d['events'])"
grep "chunk 1/1" /tmp/err.txt | tail -1 | sed 's/.*sweeps/sweeps/' | cut -c1-90
done
done
