#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "polish_matches_oracle or config2" 2>&1 | grep -v amdgpu | tail -2
for b in 2.5e8 3.75e8 7.5e8 1.5e9 3e9; do
for t in 0 1; do
NTEDIT_HIP_DEBUG=1 python bench.py --bases $b --steps 4 --warmup 2 --no-regions --no-cpu-baseline --no-gather --tune no_lane_levelling=$t 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no_levelling=$t bases=$b', 'value', d['value'], 'ms', d['ms_per_step'], d['phases_ms'])"
grep "chunk 1/1" /tmp/err.txt | tail -1 | sed 's/.*sweeps/sweeps/' | cut -c1-70
done
done
