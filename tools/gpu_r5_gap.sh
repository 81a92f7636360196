#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== this build"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "scaffold_gap" 2>&1 | grep -v amdgpu | tail -6
echo "== the build before (HEAD 579281e)"
NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_before_gap.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -s -m gpu -k "scaffold_gap" 2>&1 | grep -v amdgpu | tail -6
for b in 3.75e8 3e9; do
NTEDIT_HIP_DEBUG=1 python bench.py --bases $b --steps 3 --warmup 2 --no-regions --no-cpu-baseline --no-gather 2>/tmp/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bases=$b', 'value', d['value'], 'ms', d['ms_per_step'], d['phases_ms'])"
grep "chunk 1/1" /tmp/err.txt | tail -1 | sed 's/.*sweeps/sweeps/' | cut -c1-70
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "polish_matches_oracle or config2 or cli" 2>&1 | grep -v amdgpu | tail -2
