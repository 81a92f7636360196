#!/bin/bash
# (written for the merged-walks build of commit 818b670) parity of the small cases, a fuzz run, then the 3 Gbp step, the shard sizes
# of a strong-scaling run and configs[2].   tools/gpu_merged.sh
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_polish_matches_oracle" 2>&1 | tail -3
timeout 400 python tests/tools/fuzz_parity.py --gpu --minutes 3 --seed 7171 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>/dev/null |
  python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('3e9', d['value'], d['ms_per_step'], d['phases_ms'])"
for b in 3.75e8 7.5e8 1.5e9; do
  timeout 300 python bench.py --bases $b --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b', d['value'], d['ms_per_step'], d['phases_ms'])"
done
